#!/usr/bin/env python
"""Run-to-run spread of the discriminator gradients with batch norm: same schedule twice (serial, side stream) and
side vs serial, 4 trials each; also the old shared-workspace placement is gone (wgrad_table)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from advoc_amd import _lib
from advoc_amd.model import AdvocSmall, Modes
dev = torch.device('cuda')
g = torch.Generator().manual_seed(9)
target = torch.rand(16, 128, 513, 1, generator=g) * 2
x = target * (0.5 + torch.rand(16, 128, 513, 1, generator=g)) - 0.1
x, target = x.to(dev), target.to(dev)
BN = int(os.environ.get('BN', '1'))


def run(side):
  os.environ['ADVOC_WGRAD_STREAM'] = '1' if side else '0'
  m = AdvocSmall(Modes.TRAIN)
  m.subseq_len = 128
  m.train_batch_size = 16
  m.use_batchnorm = bool(BN)
  m.build(batch_size=16, seed=4)
  m((x, target))
  m.train_loop()
  torch.cuda.synchronize()
  st = m._built
  return {k: v.detach().clone() for k, v in st['d_G'].items()}


def rel(a, b):
  return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


S = [run(0) for _ in range(4)]
P = [run(1) for _ in range(4)]
for k in S[0]:
  ss = max(rel(S[i][k], S[0][k]) for i in range(1, 4))
  pp = max(rel(P[i][k], P[0][k]) for i in range(1, 4))
  sp = max(rel(P[i][k], S[0][k]) for i in range(4))
  print('%-55s |g| %.2e  serial rerun %.2e | side rerun %.2e | side vs serial %.2e' % (k, float(S[0][k].norm()), ss, pp, sp))
