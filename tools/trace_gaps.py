#!/usr/bin/env python
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV (one stream, serial schedule): how much of the
wall clock of a step is launch gaps rather than kernels.   python tools/trace_gaps.py DIR/*_kernel_trace.csv"""
import csv
import sys
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))))
# the timed region: from the first to the last adam_kernel launch
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[2]]
lo, hi = adam[1], adam[-1]          # skip the first D update (warm-up edge)
busy = gaps = 0
hist = {}
last_end = rows[lo][1]
for s, e, n in rows[lo + 1:hi + 1]:
  g = s - last_end
  if g > 0:
    gaps += g
    b = min(int(g / 1000) // 2 * 2, 40)
    hist[b] = hist.get(b, 0) + 1
  busy += e - max(s, last_end) if e > last_end else 0
  last_end = max(last_end, e)
span = rows[hi][1] - rows[lo][1]
n_adam = len(adam) - 1
print('%d kernels between the 2nd and the last adam launch (%d optimizer updates): span %.2f ms, kernels %.2f ms, gaps %.2f ms (%.1f %%)'
      % (hi - lo, n_adam, span / 1e6, busy / 1e6, gaps / 1e6, 100.0 * gaps / span))
print('gap histogram (us: count):', ' '.join('%d-%d:%d' % (b, b + 2, c) for b, c in sorted(hist.items())))
