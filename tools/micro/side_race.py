#!/usr/bin/env python
"""Which discriminator gradients differ between side-stream / serial execution and image / register-split kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from advoc_amd import _lib
from advoc_amd.model import AdvocSmall, Modes
dev = torch.device('cuda')
g = torch.Generator().manual_seed(9)
target = torch.rand(16, 128, 513, 1, generator=g)
x = (target * 0.7 + 0.1)
x, target = x.to(dev), target.to(dev)


def run(side, h3):
  os.environ['ADVOC_WGRAD_STREAM'] = '1' if side else '0'
  os.environ['ADVOC_H3'] = '1' if h3 else '0'
  _lib.reload_env()
  m = AdvocSmall(Modes.TRAIN)
  m.subseq_len = 128
  m.train_batch_size = 16
  m.build(batch_size=16, seed=4)
  m((x, target))
  m.train_loop()
  torch.cuda.synchronize()
  st = m._built
  return {k: v.detach().clone() for k, v in st['d_G'].items()}


def rel(a, b):
  return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


R = {(s, h): run(s, h) for s in (0, 1) for h in (0, 1)}
R2 = {(1, 1): run(1, 1), (0, 1): run(0, 1)}
for k in R[(0, 0)]:
  print('%-40s serial h3 vs x6 %.2e | side h3 vs serial h3 %.2e | side x6 vs serial x6 %.2e | side h3 rerun %.2e | serial h3 rerun %.2e' % (
      k, rel(R[(0, 1)][k], R[(0, 0)][k]), rel(R[(1, 1)][k], R[(0, 1)][k]), rel(R[(1, 0)][k], R[(0, 0)][k]),
      rel(R2[(1, 1)][k], R[(1, 1)][k]), rel(R2[(0, 1)][k], R[(0, 1)][k])))
