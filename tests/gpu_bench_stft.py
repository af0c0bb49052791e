"""Quick STFT kernel timing (not a pytest): python tests/gpu_bench_stft.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advoc_amd import spectral
for B in (32, 64, 512):
  x = (torch.rand(B, 66304, 1, 1, device='cuda') - 0.5)
  for _ in range(5):
    m = spectral.stft_magnitude(x, 1024, 256)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  n = 50
  for _ in range(n):
    m = spectral.stft_magnitude(x, 1024, 256)
  e1.record(); torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / n
  T = m.shape[1]
  byt = B * (66304 * 4 + T * 513 * 4)
  print('B=%d T=%d  %.1f us  %.2f TB/s algorithmic  %.1f Mframes/s' % (B, T, ms * 1e3, byt / ms / 1e9, B * T / ms / 1e3))
