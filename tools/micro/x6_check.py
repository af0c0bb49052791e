#!/usr/bin/env python
"""Split-bf16 gather GEMM (ADVOC_IGEMM_X6=1) against the fp32 MFMA path: error vs a float64 reference and time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from advoc_amd import conv
dev = torch.device('cuda')

def run(B, h, w, cin, cout, x6, ref=None, reps=20):
  os.environ['ADVOC_IGEMM_X6'] = '1' if x6 else '0'
  torch.manual_seed(0)
  x = torch.randn(B, h, w, cin, device=dev)
  wt = torch.randn(4, 4, cin, cout, device=dev) * 0.05
  b = torch.randn(cout, device=dev) * 0.1
  oh, ow = (h + 1) // 2, (w + 1) // 2
  y = torch.empty(B, oh, ow, cout, device=dev)
  L = conv.Layer(conv.CONV, x, y, wt, b, stride=(2, 2), pad=(conv.same_pad(h, 4, 2)[0], conv.same_pad(w, 4, 2)[0]),
                 in_act=conv.ACT_LRELU)
  for _ in range(3): L.forward()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps): L.forward()
  e1.record(); torch.cuda.synchronize()
  us = e0.elapsed_time(e1) / reps * 1e3
  dy = torch.randn_like(y)
  dx = torch.empty_like(x)
  for _ in range(3): L.backward_data(dy, dx)
  e0.record()
  for _ in range(reps): L.backward_data(dy, dx)
  e1.record(); torch.cuda.synchronize()
  us_b = e0.elapsed_time(e1) / reps * 1e3
  return y.double().cpu(), dx.double().cpu(), us, us_b, L.kernel_name(0), L.kernel_name(1), L.flops, (x, wt, b, dy)

def ref64(x, wt, b, dy, h, w):
  xd = torch.nn.functional.leaky_relu(x.double().cpu(), 0.2).requires_grad_(False)
  x0 = x.double().cpu().requires_grad_(True)
  a = torch.nn.functional.leaky_relu(x0, 0.2).permute(0, 3, 1, 2)
  pt, pb = conv.same_pad(h, 4, 2); pl, pr = conv.same_pad(w, 4, 2)
  a = torch.nn.functional.pad(a, (pl, pr, pt, pb))
  y = torch.nn.functional.conv2d(a, wt.double().cpu().permute(3, 2, 0, 1), b.double().cpu(), stride=2).permute(0, 2, 3, 1)
  (g,) = torch.autograd.grad(y, [x0], dy.double().cpu())
  return y.detach(), g

rel = lambda a, b: float((a - b).norm() / b.norm())
for (B, h, w, cin, cout) in ((8, 32, 65, 128, 256), (32, 32, 65, 128, 256), (32, 64, 129, 64, 128), (64, 32, 65, 256, 512), (32, 16, 33, 256, 256)):
  y0, d0, us0, ub0, n0, nb0, fl, t = run(B, h, w, cin, cout, False)
  y1, d1, us1, ub1, n1, nb1, _, _ = run(B, h, w, cin, cout, True)
  line = 'x[%d,%d,%d,%d]->%d  fwd %7.1f us (%5.1f TF) -> %7.1f us (%5.1f TF)  bwdD %7.1f -> %7.1f us' % (
      B, h, w, cin, cout, us0, fl / us0 / 1e6, us1, fl / us1 / 1e6, ub0, ub1)
  if B <= 8:
    yr, dr = ref64(*t, h, w)
    line += '  err fwd %.1e / %.1e  bwdD %.1e / %.1e' % (rel(y0, yr), rel(y1, yr), rel(d0, dr), rel(d1, dr))
  else:
    line += '  x6 vs fp32: fwd %.1e bwdD %.1e' % (rel(y1, y0), rel(d1, d0))
  print(line); print('    ', n0, '|', n1, '|', nb0, '|', nb1)
