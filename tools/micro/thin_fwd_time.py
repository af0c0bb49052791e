#!/usr/bin/env python
"""The 1-2 channel forward layers of AdVoc-full at B = 64 alone (no consumer images): discriminator layer_1 at 2B and B,
generator encoder_1.  python tools/micro/thin_fwd_time.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from advoc_amd import conv
dev = torch.device('cuda')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
def layer(B, c0, c1, cout, oh, ow, pad):
  x0 = torch.randn(B, 256, 513, c0, device=dev)
  x1 = torch.randn(B, 256, 513, c1, device=dev) if c1 else None
  w = torch.randn(4, 4, c0 + c1, cout, device=dev) * 0.05
  y = torch.empty(B, oh, ow, cout, device=dev)
  return conv.Layer(0, x0, y, w, torch.zeros(cout, device=dev), x1=x1, in_w=513, stride=(2, 2), pad=pad, in_act=0), y
for name, args in (('layer_1 fwd 2B', (128, 1, 1, 64, 128, 256, (1, 1))), ('layer_1 fwd B', (64, 1, 1, 64, 128, 256, (1, 1))),
                   ('encoder_1 fwd', (64, 1, 0, 64, 128, 257, (1, 1)))):
  L, y = layer(*args)
  ms = bench.event_timed(torch, L.forward, reps)
  print('%-16s %-36s %6.0f us  %.2f TB/s written' % (name, L.kernel_name(0), ms * 1e3, y.numel() * 4 / ms / 1e9), flush=True)
  del L, y
