#!/usr/bin/env python
"""Round-4 race hunt, part 4: the exact shape of the wrong elements inside the victim's tiles (AdvocSmall layer_4 backward-data,
gather_gemm_h3_kernel<2,1,2,2>: tiles of 128 grid points x 64 channels)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
sys.argv = sys.argv[:1]
os.environ['NO_CAPTURE'] = '1'
import importlib
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
src = open(os.path.join(here, 'side_race_r4b.py')).read().split("ref = run(False)")[0]
exec(src)

refc = run(False, capture=True)
lay = None
shown = 0
for t in range(40):
  r = run(True, capture=True)
  a, b = r['L3.dx'], refc['L3.dx']
  if torch.equal(a, b):
    continue
  shown += 1
  d = (a.double() - b.double())
  nz = (d != 0).nonzero()
  n_, h_, w_, c_ = a.shape
  pix = (nz[:, 0] * h_ + nz[:, 1]) * w_ + nz[:, 2]
  mt, row = pix // 128, pix % 128
  ntile, col = nz[:, 3] // 64, nz[:, 3] % 64
  print('trial %d: %d wrong elements, shape %s' % (t, len(nz), tuple(a.shape)))
  tiles = sorted(set(zip(mt.tolist(), ntile.tolist())))
  for (m, n) in tiles[:8]:
    sel = (mt == m) & (ntile == n)
    rows = sorted(set(row[sel].tolist()))
    cols = sorted(set(col[sel].tolist()))
    vals = d[tuple(nz[sel][:, k] for k in range(4))]
    refv = b.double()[tuple(nz[sel][:, k] for k in range(4))]
    print('  tile (m %d, n %d): %d wrong; rows(%d) %s; cols(%d) %s' % (m, n, int(sel.sum()), len(rows), rows[:40], len(cols), cols[:64]))
    print('     diffs: min %.3e max %.3e distinct %d; first %s' % (float(vals.min()), float(vals.max()), len(set(vals.tolist())), [float('%.3e' % v) for v in vals[:6].tolist()]))
    print('     rel to ref value: median |d/ref| %.2e' % float((vals.abs() / refv.abs().clamp_min(1e-30)).median()))
  if shown >= 4:
    break
