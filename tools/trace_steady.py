#!/usr/bin/env python
"""Per-kernel census of the STEADY-STATE train step from a rocprofv3 --kernel-trace CSV: launches and time per train_loop between
the generator's Adam launches of the last N loops (build, first-use image passes and warm-up excluded -- the whole-trace table
of trace_summary.py mixes them in).   python tools/trace_steady.py DIR/*_kernel_trace.csv [loops=2] [sequence.txt]
With a third argument the launches of the LAST loop are also written in launch order (start offset, duration, gap to the previous
launch's end, grid, kernel): which image pass / refit check / fill sits between which two layer kernels."""
import csv
import re
import sys
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Grid_Size_X', r.get('Grid_Size', '')),
                r.get('Workgroup_Size_X', r.get('Workgroup_Size', ''))) for r in csv.DictReader(open(sys.argv[1]))))
loops = int(sys.argv[2]) if len(sys.argv) > 2 else 2
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[2]]
# two Adam launches per train_loop (D, then G): the window ends behind the last G update and starts behind the G update `loops` earlier
hi = adam[-1]
lo = adam[-1 - 2 * loops]
agg = {}
def short(n):
  n = re.sub(r'^void ', '', n)
  n = n.replace('(anonymous namespace)::', '')
  n = re.sub(r'advoc::\(anonymous namespace\)::|advoc::', '', n)
  return re.sub(r'\(.*$', '', n)


for s, e, n, _, _ in rows[lo + 1:hi + 1]:
  n = short(n)
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
if len(sys.argv) > 3:
  first = adam[-3] + 1
  with open(sys.argv[3], 'w') as f:
    f.write('# last train_loop in launch order: start us | duration us | gap to previous end us | grid x / workgroup | kernel\n')
    t0, prev = rows[first][0], rows[first][0]
    for s, e, n, g, w in rows[first:hi + 1]:
      f.write('%9.1f %8.1f %6.1f  %8s/%-4s %s\n' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, g, w, short(n)[:100]))
      prev = e
