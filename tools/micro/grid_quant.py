#!/usr/bin/env python
"""Does workgroup-count quantisation (blocks per CU) show in the gather-GEMM time?
Times encoder_4 / encoder_3 (AdVoc-small) forward for neighbouring batch sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from advoc_amd import conv

dev = torch.device('cuda')
def bench(B, h, w, cin, cout, reps=30):
  oh, ow = (h + 1) // 2, (w + 1) // 2
  x = torch.randn(B, h, w, cin, device=dev)
  y = torch.empty(B, oh, ow, cout, device=dev)
  wt = torch.randn(4, 4, cin, cout, device=dev) * 0.02
  b = torch.zeros(cout, device=dev)
  l = conv.Layer(conv.CONV, x, y, wt, b, stride=(2, 2), pad=(conv.same_pad(h, 4, 2)[0], conv.same_pad(w, 4, 2)[0]),
                 in_act=conv.ACT_LRELU)
  for _ in range(5): l.forward()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps): l.forward()
  e1.record(); torch.cuda.synchronize()
  us = e0.elapsed_time(e1) / reps * 1e3
  M = B * oh * ow
  return us, l.flops / us / 1e6, l.kernel_name(0), M

if __name__ == '__main__':
 for (h, w, cin, cout) in ((32, 65, 128, 256), (64, 129, 64, 128), (16, 33, 256, 256)):
   print('layer x[B,%d,%d,%d] -> %d' % (h, w, cin, cout))
   for B in (31, 32, 33, 36, 40, 64):
     us, tf, name, M = bench(B, h, w, cin, cout)
     bm = 64 if '<1, 1, 2, 2' in name else 128
     bn = 64 if ('<1, 1, 2, 2' in name or '<2, 1,' in name) else 128
     blocks = -(-M // bm) * (cout // bn)
     print('  B=%3d  %8.1f us  %6.1f TFLOP/s  blocks=%5d (%.3f per CU)  %s' % (B, us, tf, blocks, blocks / 256., name))
