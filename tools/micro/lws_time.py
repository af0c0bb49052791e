#!/usr/bin/env python
"""Where the LWS time goes: the time-ordered (causal) pass vs the 100 batch sweeps, for 64 clips of 256 frames."""
import os, sys, ctypes, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from advoc_amd import _lib, spectral as S
lib = _lib.load()
for clips in (64, 256):
  mag = torch.rand(clips, 256, 513, device='cuda') + 0.01
  for _ in range(2):
    S.lws_spectrogram_batch(mag, 1024, 256)
  torch.cuda.synchronize()
  def timed(fn, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
      fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
  full = timed(lambda: S.lws_spectrogram_batch(mag, 1024, 256))
  nob = timed(lambda: S.lws_spectrogram_batch(mag, 1024, 256, batch=(0, 1.0, 0.0, 1.0)))
  print('%3d clips: full %.2f ms, without the batch sweeps %.2f ms, 100 sweeps %.2f ms (%.3f each)' % (
      clips, full, nob, full - nob, (full - nob) / 100), flush=True)
