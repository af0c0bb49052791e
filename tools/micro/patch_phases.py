#!/usr/bin/env python
"""The patch kernel's K loop and epilogue apart (ADVOC_H3_PATCH_ABLATE: 64 = no epilogue, 3 = no DMA and no MFMA, i.e. fragment
reads + barriers + the epilogue, 7 = the same without the barriers) on the model's shapes, no image consumers.
    python tools/micro/patch_phases.py [shape:dir ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from h3_sweep_shapes import build, setenv, timed_us

for spec in (sys.argv[1:] or ['enc2m:f', 'enc2m:d', 'dec2m:f', 'dec2m:d', 'enc3m:f', 'd4:f']):
  name, _, d = spec.partition(':')
  L, dy, dx0, dx1 = build(name)
  fn = L.forward if d != 'd' else (lambda: L.backward_data(dy, dx0, dx1))
  row = []
  for abl in [int(a) for a in os.environ.get('ABLS', '0,64,3,7,0').split(',')]:
    setenv(ADVOC_H3_PATCH=1, ADVOC_H3_PATCH_ABLATE=0, ADVOC_H3_SKIP_PREP=None)
    fn()
    setenv(ADVOC_H3_SKIP_PREP=1, ADVOC_H3_PATCH_ABLATE=abl)
    us = timed_us(fn, 10)
    row.append('abl %3d %7.1f us' % (abl, us))
  setenv(ADVOC_H3_SKIP_PREP=None, ADVOC_H3_PATCH_ABLATE=0)
  print('%-6s %s %-28s %s' % (name, d, L.kernel_name(1 if d == 'd' else 0), ' | '.join(row)), flush=True)
  del L, dy, dx0, dx1
  torch.cuda.empty_cache()
