#!/usr/bin/env python
"""The fused extractor (csrc/extract.hip) against the two launches it replaces, at 512 clips and at the training feed."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from advoc_amd import spectral
from advoc_amd.spectral_util import SpectralUtil
su = SpectralUtil()
wav = bench.synth_waveforms(64, 1, torch.device('cuda'))
r = bench.extractor_leg(torch, spectral, su, wav)
print('ADVOC_EXTRACT_WAVES=%s' % os.environ.get('ADVOC_EXTRACT_WAVES', '(default)'))
print(json.dumps(r['triple'], indent=1))
