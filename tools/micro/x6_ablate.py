#!/usr/bin/env python
"""Times one large split-bf16 layer (forward + backward-data); used with ablation builds of igemm.hip."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from advoc_amd import conv
dev = torch.device('cuda')
B, h, w, cin, cout = 64, 32, 65, 256, 512
x = torch.randn(B, h, w, cin, device=dev); wt = torch.randn(4, 4, cin, cout, device=dev) * 0.05
y = torch.empty(B, 16, 33, cout, device=dev)
L = conv.Layer(conv.CONV, x, y, wt, None, stride=(2, 2), pad=(1, 1), in_act=conv.ACT_LRELU)
dy = torch.randn_like(y); dx = torch.empty_like(x)
def t(fn, reps=20):
  for _ in range(3): fn()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e3
print('%s  fwd %.1f us (%.1f TF)  bwdD %.1f us' % (sys.argv[1] if len(sys.argv) > 1 else '', t(L.forward), L.flops / t(L.forward) / 1e6,
                                                   t(lambda: L.backward_data(dy, dx))), L.kernel_name(0))
