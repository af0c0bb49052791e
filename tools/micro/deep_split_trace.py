#!/usr/bin/env python
"""Kernel-only times of the deep layers' gather GEMMs under a forced number of workspace K slices: host overhead of a Python
call is 40-60 us, more than these kernels take, so the times come from a rocprofv3 kernel trace.
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/micro/deep_split_trace.py run CONFIGS.txt [shape ...]
    python tools/micro/deep_split_trace.py table CONFIGS.txt DIR/*/*_kernel_trace.csv
`run` launches every configuration R times in a fixed order and writes that order; `table` cuts the trace into groups of R."""
import os, sys
R = 8
SPLITS = [0, 2, 3, 4, 5, 6, 8, 10, 12, 16]
MATCH = 'gather_gemm_h3_kernel'

if sys.argv[1] == 'run':
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  import torch
  from h3_sweep_shapes import build, setenv
  out = open(sys.argv[2], 'w')
  for name in (sys.argv[3:] or ['enc5m', 'enc6m', 'enc7m', 'enc8m', 'dec5m', 'dec6m', 'dec7m', 'dec8m']):
    for sp in SPLITS:
      setenv(ADVOC_H3_DEEP_SPLIT=sp or None, ADVOC_H3_SKIP_PREP=None, ADVOC_H3_TILE=os.environ.get('TILE') or None)
      L, dy, dx0, dx1 = build(name)          # (a fresh layer: the workspace is sized under the setting)
      for d, tag in ((0, 'fwd'), (1, 'bwdD')):
        fn = L.forward if d == 0 else (lambda: L.backward_data(dy, dx0, dx1))
        fn()                                  # operand images
        torch.cuda.synchronize()
        setenv(ADVOC_H3_SKIP_PREP=1)
        for _ in range(R):
          fn()
        torch.cuda.synchronize()
        setenv(ADVOC_H3_SKIP_PREP=None)
        out.write('%s %s %d %s\n' % (name, tag, sp, L.kernel_name(d)))
      del L, dy, dx0, dx1
      torch.cuda.empty_cache()
  out.close()
else:
  import csv
  cfgs = [l.split(None, 3) for l in open(sys.argv[2])]
  rows = [r for r in csv.DictReader(open(sys.argv[3])) if MATCH in r['Kernel_Name']]
  rows.sort(key=lambda r: int(r['Start_Timestamp']))
  per = R + 1
  assert len(rows) == per * len(cfgs), (len(rows), len(cfgs))
  table = {}
  for i, (name, tag, sp, kern) in enumerate(cfgs):
    d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows[i * per + 1:(i + 1) * per]]
    g = rows[i * per + 1].get('Grid_Size') or rows[i * per + 1].get('Grid_Size_X')
    table.setdefault((name, tag), {})[int(sp)] = (sum(d) / len(d) / 1e3, int(g) // 256)
  print('shape  dir  | ' + ' '.join('%12s' % ('dflt' if s == 0 else 's=%d' % s) for s in SPLITS) + '   (us, workgroups)')
  for (name, tag), t in table.items():
    print('%-6s %-4s | ' % (name, tag) + ' '.join('%6.1f %5d' % t[s] for s in SPLITS))
