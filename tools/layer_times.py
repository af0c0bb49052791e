#!/usr/bin/env python
"""Per-layer / per-direction kernel timing of one AdVoc train_loop (HIP events).
    python tools/layer_times.py [small|regular] [batch]"""
import os, sys
os.environ['ADVOC_WGRAD_STREAM'] = '0'   # per-kernel times: no concurrent side-stream kernels
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advoc_amd import conv
from advoc_amd.model import Advoc, AdvocSmall, Modes

kind = sys.argv[1] if len(sys.argv) > 1 else 'small'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
m = (AdvocSmall if kind == 'small' else Advoc)(Modes.TRAIN)
m.build(batch_size=B)
dev = torch.device('cuda')
x = torch.rand(B, 256, 513, 1, device=dev)
t = torch.rand(B, 256, 513, 1, device=dev)
m((x, t))
for _ in range(2):
  m.train_loop()
torch.cuda.synchronize()

recs = []
orig = conv.Layer._run
def run(self, direction, fn, **kw):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record(); fn(); e1.record()
  recs.append((self, direction, e0, e1))
conv.Layer._run = run
N = 3
for _ in range(N):
  m.train_loop()
torch.cuda.synchronize()
conv.Layer._run = orig
names = {}
st = m._built
for k, l in st['g_layers'].items(): names[id(l)] = 'G.' + k
for i, l in enumerate(st['d_layers_2b']): names[id(l)] = 'D2b.layer_%d' % (i + 1)
for i, l in enumerate(st['d_layers_fake']): names[id(l)] = 'Df.layer_%d' % (i + 1)
agg = {}
for l, d, e0, e1 in recs:
  key = (names[id(l)], d)
  a = agg.setdefault(key, [0, 0.0, l])
  a[0] += 1; a[1] += e0.elapsed_time(e1)
tot = sum(a[1] for a in agg.values())
print('%-22s %-4s %-38s %6s %9s %8s  shape' % ('layer', 'dir', 'kernel', 'calls', 'us/call', 'TFLOP/s'))
for (name, d), (n, ms, l) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
  us = ms / n * 1e3
  s = l.struct
  shape = 'x[%d,%d,%d,%d+%d] y[%d,%d,%d]' % (s.x0.n, s.x0.h, s.x0.w, s.x0.c, s.x1.c if s.x1.p else 0, s.y.h, s.y.w, s.y.c)
  print('%-22s %-4s %-38s %6d %9.1f %8.2f  %s' % (name, ['fwd', 'bwdD', 'bwdW'][d], l.kernel_name(d), n, us, l.flops / us / 1e6, shape))
print('total %.2f ms per train_loop' % (tot / N))
