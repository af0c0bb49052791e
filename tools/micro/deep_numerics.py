#!/usr/bin/env python
"""Error of the deep layers' gather GEMMs against float64 (forward and backward-data) under the two tile / K-slice plans
(ADVOC_H3_DEEP_PLAN=0|1), and the largest difference between the two plans' results.
    python tools/micro/deep_numerics.py [shape ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import torch.nn.functional as F
from h3_sweep_shapes import SHAPES, setenv
from advoc_amd import conv
dev = torch.device('cuda')


def rel(a, b):
  return float((a.double() - b).norm() / b.norm())


for name in (sys.argv[1:] or ['enc5m', 'enc6m', 'enc7m', 'enc8m', 'dec5m', 'dec6m', 'dec7m', 'dec8m']):
  kind, B, H, W, c0, c1, cout, stride, trim, act = SHAPES[name]
  g = torch.Generator().manual_seed(1)
  x0 = (torch.randn(B, H, W + trim, c0, generator=g) * torch.exp(torch.randn(B, H, W + trim, c0, generator=g))).to(dev)
  x1 = torch.randn(B, H, W, c1, generator=g).to(dev) if c1 else None
  w = (torch.randn(4, 4, *((c0 + c1, cout) if kind == 0 else (cout, c0 + c1)), generator=g) * 0.02).to(dev)
  oh, ow = (-(-H // 2), -(-W // 2)) if kind == 0 else (2 * H, 2 * W)
  dy = (torch.randn(B, oh, ow, cout, generator=g) * 1e-3).to(dev)
  xin = torch.cat([x0[:, :, :W]] + ([x1] if c1 else []), dim=3).double()
  a = (F.leaky_relu(xin, 0.2) if act == 1 else torch.relu(xin)).requires_grad_(True)
  if kind == 0:
    ap = F.pad(a.permute(0, 3, 1, 2), (1, 2 if W % 2 else 1, 1, 2 if H % 2 else 1))
    y64 = F.conv2d(ap, w.double().permute(3, 2, 0, 1), stride=stride).permute(0, 2, 3, 1)[:, :oh, :ow]
  else:
    y64 = F.conv_transpose2d(a.permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), stride=2, padding=1).permute(0, 2, 3, 1)
  (ga,) = torch.autograd.grad(y64, a, dy.double())
  res = {}
  for plan in (0, 1):
    setenv(ADVOC_H3_DEEP_PLAN=plan)
    y = torch.empty(B, oh, ow, cout, device=dev)
    L = conv.Layer(kind, x0, y, w, None, x1=x1, in_w=W, stride=stride, pad=(1, 1), in_act=act)
    L.forward()
    dx0 = torch.zeros_like(x0)
    dx1 = torch.zeros_like(x1) if c1 else None
    L.struct.in_act = 0            # backward-data WITHOUT the activation gate: d / d(act(x))
    L.backward_data(dy, dx0, dx1)
    gx = torch.cat([dx0[:, :, :W]] + ([dx1] if c1 else []), dim=3)
    torch.cuda.synchronize()
    res[plan] = (y.clone(), gx.clone(), L.kernel_name(0), L.kernel_name(1))
  setenv(ADVOC_H3_DEEP_PLAN=None)
  print('%-6s fwd  plan0 %.3e (%s)  plan1 %.3e (%s)  max|diff| / max|y| %.2e' % (
      name, rel(res[0][0], y64.detach()), res[0][2][-12:], rel(res[1][0], y64.detach()), res[1][2][-12:],
      float((res[0][0] - res[1][0]).abs().max() / y64.abs().max())))
  print('%-6s bwdD plan0 %.3e (%s)  plan1 %.3e (%s)  max|diff| / max|dx| %.2e' % (
      name, rel(res[0][1], ga), res[0][3][-12:], rel(res[1][1], ga), res[1][3][-12:],
      float((res[0][1] - res[1][1]).abs().max() / ga.abs().max())), flush=True)
