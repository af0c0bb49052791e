#!/usr/bin/env python
"""Where the patch kernel's time goes: ADVOC_H3_PATCH_ABLATE bits (1 no DMA, 2 no MFMA, 4 no barrier) on one shape.
    python tools/micro/patch_ablate.py [shape:dir ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from h3_sweep_shapes import build, setenv, timed_us

for spec in (sys.argv[1:] or ['d4:f', 'dec2m:f', 'dec4m:f', 'enc2m:d']):
  name, _, d = spec.partition(':')
  L, dy, dx0, dx1 = build(name)
  fn = L.forward if d != 'd' else (lambda: L.backward_data(dy, dx0, dx1))
  for waves in (8,):
    row = []
    for abl in (0, 8, 16, 0, 8, 16):
      setenv(ADVOC_H3_PATCH=1, ADVOC_H3_PATCH_ABLATE=0, ADVOC_H3_SKIP_PREP=None)
      fn()
      setenv(ADVOC_H3_SKIP_PREP=1, ADVOC_H3_PATCH_ABLATE=abl)
      us = timed_us(fn, 5)
      row.append('abl%d %7.1f us %5.1f TF' % (abl, us, L.flops / us / 1e6))
    setenv(ADVOC_H3_SKIP_PREP=None, ADVOC_H3_PATCH_ABLATE=0)
    print('%-6s %s w%d %s | %s' % (name, d, waves, L.kernel_name(1 if d == 'd' else 0), ' | '.join(row)), flush=True)
  del L, dy, dx0, dx1
  torch.cuda.empty_cache()
