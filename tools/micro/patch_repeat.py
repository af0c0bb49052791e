#!/usr/bin/env python
"""Bitwise repeatability of the patch kernels under a busy second stream: every launch of a deterministic kernel on the same
inputs must give the same bits; a difference is a race (LDS-DMA ring, early prologue, scratch reuse).
    python tools/micro/patch_repeat.py [reps=40] [shape:dir ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from h3_sweep_shapes import build

args = sys.argv[1:]
reps = int(args[0]) if args and args[0].isdigit() else 40
specs = [a for a in args if not a.isdigit()] or ['enc2m:f', 'enc2m:d', 'dec2m:f', 'dec2m:d', 'enc3m:f', 'enc3m:d', 'd4b:f', 'd4b:d', 'dec3m:f']
side = torch.cuda.Stream()
junk = torch.randn(64 * 1024 * 1024, device='cuda')
bad_total = 0
for spec in specs:
  name, _, d = spec.partition(':')
  L, dy, dx0, dx1 = build(name)
  outs = [L.y] if d != 'd' else [t for t in (dx0, dx1) if t is not None]
  fn = L.forward if d != 'd' else (lambda: L.backward_data(dy, dx0, dx1))
  fn()
  for o in outs:
    o.fill_(float('nan'))          # (elements a launch does not own -- a trimmed column -- compare as what the fill left)
  fn()
  torch.cuda.synchronize()
  ref = [o.clone() for o in outs]
  bad = 0
  for r in range(reps):
    if r % 2:
      with torch.cuda.stream(side):          # an HBM-bound stranger beside every other launch
        junk.mul_(1.0000001)
    for o in outs:
      o.fill_(float('nan'))
    fn()
    torch.cuda.synchronize()
    for o, q in zip(outs, ref):
      if not torch.equal(o.view(torch.int32), q.view(torch.int32)):
        bad += 1
        n = int((o.view(torch.int32) != q.view(torch.int32)).sum())
        print('  %s rep %d: %d elements differ' % (spec, r, n), flush=True)
  print('%-8s %-28s %d launches, %d differing' % (spec, L.kernel_name(1 if d == 'd' else 0), reps, bad), flush=True)
  bad_total += bad
  del L, dy, dx0, dx1, outs, ref
  torch.cuda.empty_cache()
print('TOTAL differing launches: %d' % bad_total)
