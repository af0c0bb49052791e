#!/usr/bin/env python
"""advoc_mel_pinv_f32 alone (preallocated outputs, HIP events) against the two advoc_matmul_nt_f32 launches it replaces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from advoc_amd import _lib, spectral
from advoc_amd.spectral_util import SpectralUtil
su = SpectralUtil()
lib = _lib.load()
for clips in (512, 128):
  rows = clips * 256
  mag = torch.rand(rows, 513, device='cuda') * 10
  mel = torch.empty(rows, 80, device='cuda')
  inv = torch.empty(rows, 513, device='cuda')
  W, P = su.meltrans, su.invmeltrans
  B, WP, PT = su._const('packed')

  def fused():
    _lib.check(lib.advoc_mel_pinv_f32(_lib.ptr(mag), _lib.ptr(WP), _lib.ptr(B), _lib.ptr(PT), _lib.ptr(mel), _lib.ptr(inv),
                                      rows, 513, 80, WP.numel(), _lib.stream()), 'mel_pinv')

  def two():
    _lib.check(lib.advoc_matmul_nt_f32(_lib.ptr(mag), _lib.ptr(W), _lib.ptr(mel), rows, 513, 80, _lib.stream()), 'a')
    _lib.check(lib.advoc_matmul_nt_f32(_lib.ptr(mel), _lib.ptr(P), _lib.ptr(inv), rows, 80, 513, _lib.stream()), 'b')
  for name, fn in (('fused', fused), ('two matmuls', two)):
    for _ in range(3):
      fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
      fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    gb = rows * (513 * 4 + 80 * 4 + 513 * 4) / 1e9
    print('%4d clips %-12s %8.1f us  %7.1f GB/s algorithmic' % (clips, name, ms * 1e3, gb / ms * 1e3), flush=True)
