#!/usr/bin/env python
"""igemm_patch.hip against the per-tap tiles of igemm_h3.hip on the stride-1 gathers of the model: GEMM kernel alone
(ADVOC_H3_SKIP_PREP=1 after the first call); the two must agree to round-off.
    python tools/micro/patch_sweep.py [shape[:dirs] ...]      dirs: f forward, d backward-data"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from h3_sweep_shapes import build, setenv, timed_us

DEFAULT = ['dec2m:f', 'dec3m:f', 'dec4m:f', 'enc2m:d', 'enc3m:d', 'enc4m:d', 'd2m:d', 'd3m:d', 'd4:fd', 'd4b:fd']
S2 = ['enc2m:f', 'enc3m:f', 'd2m:f', 'd3m:f', 'dec2m:d', 'dec3m:d', 'enc4m:f', 'dec4m:d']     # stride-2 gathers (parity planes)


def rel(a, b):
  return float((a.double() - b.double()).norm() / b.double().norm())


for spec in (S2 if sys.argv[1:] == ['s2'] else (sys.argv[1:] or DEFAULT)):
  name, _, dirs = spec.partition(':')
  dirs = dirs or 'fd'
  L, dy, dx0, dx1 = build(name)
  for d, tag in ((0, 'fwd '), (1, 'bwdD')):
    if 'fd'[d] not in dirs:
      continue
    fn = L.forward if d == 0 else (lambda: L.backward_data(dy, dx0, dx1))
    out = (lambda: L.y) if d == 0 else (lambda: dx0)
    res = []
    for patch, persist in ((0, 1), (1, 0), (1, 1)):
      setenv(ADVOC_H3_PATCH=patch, ADVOC_H3_PATCH_PERSIST=persist, ADVOC_H3_SKIP_PREP=None)
      L._names = {}
      fn()
      ref = out().clone()
      setenv(ADVOC_H3_SKIP_PREP=1)
      us = timed_us(fn)
      setenv(ADVOC_H3_SKIP_PREP=None)
      res.append((L.kernel_name(d), us, ref, 'persist%d' % persist))
    n0, u0, r0, _ = res[0]
    line = '%-6s %s %-34s %7.1f us %5.1f TF |' % (name, tag, n0, u0, L.flops / u0 / 1e6)
    for n1, u1, r1, vt in res[1:]:
      line += ' %s %7.1f us %5.1f TF x%.2f%s |' % (vt, u1, L.flops / u1 / 1e6, u0 / u1, '' if rel(r1, r0) < 1e-6 else ' MISMATCH %.1e' % rel(r1, r0))
    print(line, flush=True)
  del L, dy, dx0, dx1
  torch.cuda.empty_cache()
