#!/usr/bin/env python
"""The deep layers' gather GEMMs (GEMM kernel alone) under the LDS-stage count of the 128 x 64 per-tap tile and the number
of workgroups per CU the workspace K split aims at.
    python tools/micro/deep_stages.py [shape ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from h3_sweep_shapes import build, setenv, timed_us

VARIANTS = [('ns2 w2', dict(ADVOC_H3_DEEP_STAGES=2, ADVOC_H3_DEEP_WGS_PER_CU=2)),
            ('ns2 w3', dict(ADVOC_H3_DEEP_STAGES=2, ADVOC_H3_DEEP_WGS_PER_CU=3)),
            ('ns3 w2', dict(ADVOC_H3_DEEP_STAGES=3, ADVOC_H3_DEEP_WGS_PER_CU=2)),
            ('ns3 w4', dict(ADVOC_H3_DEEP_STAGES=3, ADVOC_H3_DEEP_WGS_PER_CU=4)),
            ('ns4 w1', dict(ADVOC_H3_DEEP_STAGES=4, ADVOC_H3_DEEP_WGS_PER_CU=1)),
            ('ns4 w2', dict(ADVOC_H3_DEEP_STAGES=4, ADVOC_H3_DEEP_WGS_PER_CU=2))]
CLEAR = dict(ADVOC_H3_DEEP_STAGES=None, ADVOC_H3_DEEP_WGS_PER_CU=None, ADVOC_H3_SKIP_PREP=None)
tot = [0.0] * len(VARIANTS)
for name in (sys.argv[1:] or ['enc5m', 'enc6m', 'enc7m', 'enc8m', 'dec5m', 'dec6m', 'dec7m', 'dec8m']):
  L, dy, dx0, dx1 = build(name)
  for d, tag in ((0, 'fwd '), (1, 'bwdD')):
    fn = L.forward if d == 0 else (lambda: L.backward_data(dy, dx0, dx1))
    line = '%-6s %s' % (name, tag)
    for i, (vt, env) in enumerate(VARIANTS):
      setenv(**dict(CLEAR, **env))
      L._names = {}
      fn()
      setenv(ADVOC_H3_SKIP_PREP=1)
      us = timed_us(fn, reps=20)
      tot[i] += us
      line += ' | %s %7.1f us %5.1f TF' % (vt, us, L.flops / us / 1e6)
    setenv(**CLEAR)
    print(line + '   ' + L.kernel_name(d), flush=True)
  del L, dy, dx0, dx1
  torch.cuda.empty_cache()
print('sum    ' + ' | '.join('%s %8.1f us' % (v[0], t) for v, t in zip(VARIANTS, tot)))
