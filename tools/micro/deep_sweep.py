#!/usr/bin/env python
"""The deep layers' gather GEMMs (GEMM kernel alone) under the split-K / tile switches.
    python tools/micro/deep_sweep.py [shape ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from h3_sweep_shapes import build, setenv, timed_us

VARIANTS = [('default', {}), ('nosplit', dict(ADVOC_IGEMM_SPLITK=0)), ('min32', dict(ADVOC_H3_MIN_TILES=32)),
            ('min8', dict(ADVOC_H3_MIN_TILES=8)), ('min1', dict(ADVOC_H3_MIN_TILES=1))]
for name in (sys.argv[1:] or ['enc5m', 'enc6m', 'enc7m', 'enc8m', 'dec5m', 'dec6m', 'dec7m', 'dec8m']):
  L, dy, dx0, dx1 = build(name)
  for d, tag in ((0, 'fwd '), (1, 'bwdD')):
    fn = L.forward if d == 0 else (lambda: L.backward_data(dy, dx0, dx1))
    line = '%-6s %s' % (name, tag)
    for vt, env in VARIANTS:
      setenv(**dict(dict(ADVOC_IGEMM_SPLITK=None, ADVOC_H3_TILE=None, ADVOC_H3_MIN_TILES=None, ADVOC_H3_SKIP_PREP=None), **env))
      L._names = {}
      fn()
      setenv(ADVOC_H3_SKIP_PREP=1)
      us = timed_us(fn)
      line += ' | %s %s %7.1f us %5.1f TF' % (vt, L.kernel_name(d)[-12:], us, L.flops / us / 1e6)
    setenv(ADVOC_IGEMM_SPLITK=None, ADVOC_H3_TILE=None, ADVOC_H3_MIN_TILES=None, ADVOC_H3_SKIP_PREP=None)
    print(line, flush=True)
  del L, dy, dx0, dx1
  torch.cuda.empty_cache()
