#!/usr/bin/env python
"""Times the gather-GEMM variants on full-model layer shapes (forward / backward-data), GEMM kernel alone
(ADVOC_H3_SKIP_PREP=1 after the first call: the operand images stay in the workspace).
    python tools/micro/h3_sweep.py [shape ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from advoc_amd import _lib, conv
dev = torch.device('cuda')
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from h3_sweep_shapes import SHAPES, build


def setenv(**kw):
  for k, v in kw.items():
    if v is None:
      os.environ.pop(k, None)
    else:
      os.environ[k] = str(v)
  _lib.reload_env()


def t(fn, reps=10):
  for _ in range(2):
    fn()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e3


VARIANTS = [('t5_256x256', dict(ADVOC_H3=1, ADVOC_H3_TILE=5)), ('t6_256x128', dict(ADVOC_H3=1, ADVOC_H3_TILE=6))] + [
    ('t%d_s%d' % (tl, st), dict(ADVOC_H3=1, ADVOC_H3_TILE=tl, ADVOC_H3_STAGES=st)) for tl, st in ((1, 2),)]

for name in (sys.argv[1:] or list(SHAPES)):
  L, dy, dx0, dx1 = build(name)
  for tag, env in VARIANTS:
    setenv(**dict(dict(ADVOC_H3_SKIP_PREP=None, ADVOC_H3_TILE=None, ADVOC_H3_STAGES=None), **env))
    row = []
    for d, fn in ((0, L.forward), (1, lambda: L.backward_data(dy, dx0, dx1))):
      L._names = {}
      fn()                                    # images + weights into the workspace
      setenv(ADVOC_H3_SKIP_PREP=1)
      us = t(fn)
      setenv(ADVOC_H3_SKIP_PREP=None)
      us_all = t(fn, 5)
      row.append('%s %-34s %8.1f us %6.1f TF (with prep %8.1f us)' % (['fwd ', 'bwdD'][d], L.kernel_name(d), us, L.flops / us / 1e6, us_all))
    print('%-5s %-6s %s | %s' % (name, tag, row[0], row[1]), flush=True)
