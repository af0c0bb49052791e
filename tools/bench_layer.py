#!/usr/bin/env python
"""Micro-benchmark of single conv-layer launches (for rocprofv3 counter runs and A/B tuning).
    python tools/bench_layer.py NAME [reps]      NAME in the table below."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advoc_amd import conv

CASES = {
    # name: (kind, B, H, W, c0, c1, cout, stride, pad, act, trim)
    'd4':    (0, 64, 32, 64, 128, 0, 256, (1, 1), (1, 1), 1, 0),
    'd3':    (0, 64, 64, 128, 64, 0, 128, (2, 2), (1, 1), 1, 0),
    'd2':    (0, 64, 128, 256, 32, 0, 64, (2, 2), (1, 1), 1, 0),
    'enc2':  (0, 32, 128, 257, 32, 0, 64, (2, 2), (1, 1), 1, 0),
    'enc3':  (0, 32, 64, 129, 64, 0, 128, (2, 2), (1, 1), 1, 0),
    'enc4':  (0, 32, 32, 65, 128, 0, 256, (2, 2), (1, 1), 1, 0),
    'enc5':  (0, 32, 16, 33, 256, 0, 256, (2, 2), (1, 1), 1, 0),
    'dec5':  (1, 32, 8, 17, 256, 0, 256, (2, 2), (1, 1), 2, 0),
    'dec4':  (1, 32, 16, 33, 256, 256, 128, (2, 2), (1, 1), 2, 1),
    'dec3':  (1, 32, 32, 65, 128, 128, 64, (2, 2), (1, 1), 2, 1),
    'dec2':  (1, 32, 64, 129, 64, 64, 32, (2, 2), (1, 1), 2, 1),
    'dec1':  (1, 32, 128, 257, 32, 32, 1, (2, 2), (1, 1), 2, 1),
    'enc1':  (0, 32, 256, 513, 1, 0, 32, (2, 2), (1, 1), 0, 0),
    'd1':    (0, 64, 256, 513, 1, 1, 32, (2, 2), (1, 1), 0, 0),
    'd5':    (0, 64, 31, 63, 256, 0, 1, (1, 1), (1, 1), 1, 0),
}

def build(name):
  kind, B, H, W, c0, c1, cout, stride, pad, act, trim = CASES[name]
  dev = torch.device('cuda')
  x0 = torch.randn(B, H, W + trim, c0, device=dev)
  x1 = torch.randn(B, H, W, c1, device=dev) if c1 else None
  cin = c0 + c1
  if kind == 0:
    oh = (H + 2 * pad[0] - 4) // stride[0] + 1 if name.startswith('d') and not name.startswith('dec') else -(-H // stride[0])
    ow = (W + 2 * pad[1] - 4) // stride[1] + 1 if name.startswith('d') and not name.startswith('dec') else -(-W // stride[1])
    w = torch.randn(4, 4, cin, cout, device=dev) * 0.05
  else:
    oh, ow = 2 * H, 2 * W - (1 if cout == 1 else 0)
    w = torch.randn(4, 4, cout, cin, device=dev) * 0.05
  y = torch.empty(B, oh, ow, cout, device=dev)
  b = torch.zeros(cout, device=dev)
  L = conv.Layer(kind, x0, y, w, b, x1=x1, in_w=W, stride=stride, pad=pad, in_act=act)
  dy = torch.randn_like(y)
  dx0 = torch.zeros_like(x0)
  dx1 = torch.zeros_like(x1) if x1 is not None else None
  dw = torch.zeros_like(w)
  return L, dy, dx0, dx1, dw

def main():
  name = sys.argv[1]
  reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
  dirs = sys.argv[3] if len(sys.argv) > 3 else 'fdw'
  L, dy, dx0, dx1, dw = build(name)
  fns = {'f': ('fwd', 0, lambda: L.forward()), 'd': ('bwdD', 1, lambda: L.backward_data(dy, dx0, dx1)),
         'w': ('bwdW', 2, lambda: L.backward_weight(dy, dw))}
  for k in dirs:
    tag, d, fn = fns[k]
    for _ in range(2):
      fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
      fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print('%-6s %-5s %-40s %9.1f us  %7.2f TFLOP/s  %7.1f GB/s(alg)' % (
        name, tag, L.kernel_name(d), us, L.flops / us / 1e6, L.bytes_fwd / us / 1e3))

if __name__ == '__main__':
  main()
