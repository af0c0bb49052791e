import sys, torch, json
sys.path.insert(0, ".")
import bench
from advoc_amd import spectral
from advoc_amd.spectral_util import SpectralUtil
su = SpectralUtil()
wav = bench.synth_waveforms(64, 1, torch.device("cuda"))
bench.extractor_leg(torch, spectral, su, wav)
r = bench.extractor_leg(torch, spectral, su, wav)
print("stft", round(r["avg_launch_ms"]*1e3,1), "triple bulk one/two", round(r["triple"]["one_launch_ms"]*1e3,1), round(r["triple"]["two_launches_ms"]*1e3,1), "feed one/two", round(r["triple"]["at_train_feed"]["one_launch_ms"]*1e3,1), round(r["triple"]["at_train_feed"]["two_launches_ms"]*1e3,1))
