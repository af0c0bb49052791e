#!/usr/bin/env python
"""The deep layers' gather GEMMs (GEMM kernel alone) under a forced number of workspace K slices (ADVOC_H3_DEEP_SPLIT).
    python tools/micro/deep_split.py [shape ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from h3_sweep_shapes import build, setenv, timed_us

SPLITS = [0, 2, 3, 4, 5, 6, 8, 10, 12, 16]
CLEAR = dict(ADVOC_H3_DEEP_SPLIT=None, ADVOC_H3_SKIP_PREP=None)
print('shape  dir  | ' + ' '.join('%7s' % ('dflt' if s == 0 else 's=%d' % s) for s in SPLITS) + '   (us)')
for name in (sys.argv[1:] or ['enc6m', 'enc7m', 'enc8m', 'dec6m', 'dec7m', 'dec8m']):
  L, dy, dx0, dx1 = build(name)
  for d, tag in ((0, 'fwd '), (1, 'bwdD')):
    fn = L.forward if d == 0 else (lambda: L.backward_data(dy, dx0, dx1))
    line = '%-6s %s |' % (name, tag)
    for sp in SPLITS:
      setenv(**dict(CLEAR, ADVOC_H3_DEEP_SPLIT=sp or None))
      L._names = {}
      fn()
      setenv(ADVOC_H3_SKIP_PREP=1)
      line += ' %7.1f' % timed_us(fn, reps=20)
    setenv(**CLEAR)
    print(line + '   ' + L.kernel_name(d), flush=True)
  del L, dy, dx0, dx1
  torch.cuda.empty_cache()
