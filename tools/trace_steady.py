#!/usr/bin/env python
"""Per-kernel census of the STEADY-STATE train step from a rocprofv3 --kernel-trace CSV: launches and time per train_loop between
the generator's Adam launches of the last N loops (build, first-use image passes and warm-up excluded -- the whole-trace table
of trace_summary.py mixes them in).   python tools/trace_steady.py DIR/*_kernel_trace.csv [loops=2]"""
import csv
import re
import sys
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))))
loops = int(sys.argv[2]) if len(sys.argv) > 2 else 2
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[2]]
# two Adam launches per train_loop (D, then G): the window ends behind the last G update and starts behind the G update `loops` earlier
hi = adam[-1]
lo = adam[-1 - 2 * loops]
agg = {}
for s, e, n in rows[lo + 1:hi + 1]:
  n = re.sub(r'^void ', '', n)
  n = n.replace('(anonymous namespace)::', '')
  n = re.sub(r'advoc::\(anonymous namespace\)::|advoc::', '', n)
  n = re.sub(r'\(.*$', '', n)
  a = agg.setdefault(n, [0, 0])
  a[0] += 1
  a[1] += e - s
span = rows[hi][1] - rows[lo][1]
tot = sum(a[1] for a in agg.values())
print('steady window: %d train_loops, span %.3f ms per loop, kernel time %.3f ms per loop, %d launches per loop'
      % (loops, span / 1e6 / loops, tot / 1e6 / loops, sum(a[0] for a in agg.values()) / loops))
print('| kernel | launches / loop | ms / loop | avg us | % |')
print('|---|---|---|---|---|')
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
  print('| `%s` | %.1f | %.3f | %.1f | %.2f |' % (n[:110], c / loops, t / 1e6 / loops, t / 1e3 / c, 100.0 * t / tot))
