#!/usr/bin/env python
"""Weight-gradient kernels on full-model layer shapes: register-split (wgrad.hip) vs operand images (wgrad_h3.hip),
kernel alone (images reused) and with the image passes.   python tools/micro/wgrad_sweep.py [shape ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from h3_sweep_shapes import SHAPES, build
from advoc_amd import _lib


def setenv(**kw):
  for k, v in kw.items():
    if v is None:
      os.environ.pop(k, None)
    else:
      os.environ[k] = str(v)
  _lib.reload_env()


def t(fn, reps=6):
  for _ in range(2):
    fn()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e3


for name in (sys.argv[1:] or list(SHAPES)):
  L, dy, dx0, dx1 = build(name)
  dw = torch.zeros_like(L.weight)
  fn = lambda: L.backward_weight(dy, dw)      # noqa: E731
  setenv(ADVOC_WGRAD_H3=0, ADVOC_H3_SKIP_PREP=None)
  L._names = {}
  us_old = t(fn)
  old = dw.clone()
  name_old = L.kernel_name(2)
  line = '%-6s %-36s %8.1f us %6.1f TF |' % (name, name_old, us_old, L.flops / us_old / 1e6)
  for tile in (1, 2):
    setenv(ADVOC_WGRAD_H3=1, ADVOC_WGRAD_H3_TILE=tile)
    L._names = {}
    us_all = t(fn)
    new = dw.clone()
    setenv(ADVOC_H3_SKIP_PREP=1)
    us_new = t(fn)
    setenv(ADVOC_H3_SKIP_PREP=None)
    rel = float((new.double() - old.double()).norm() / old.double().norm())
    line += ' %-20s %8.1f us %6.1f TF (with images %8.1f us) rel %.1e |' % (L.kernel_name(2), us_new, L.flops / us_new / 1e6, us_all, rel)
  setenv(ADVOC_WGRAD_H3_TILE=None)
  print(line, flush=True)
  del L, dy, dx0, dx1, dw
  torch.cuda.empty_cache()
