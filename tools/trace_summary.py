#!/usr/bin/env python
"""Per-kernel table from a rocprofv3 --kernel-trace --output-format csv run.
    python tools/trace_summary.py DIR/*_kernel_trace.csv [--skip-first N] > profiles/rNN_x_kernel_trace.md
--skip-first drops the first N dispatches of every kernel (warm-up iterations).
--by-grid SUBSTR adds one row per grid size for the kernels whose name contains SUBSTR (which launches of a shared instance
cost what)."""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_summary import short


def main():
  path = sys.argv[1]
  skip = int(sys.argv[sys.argv.index('--skip-first') + 1]) if '--skip-first' in sys.argv else 0
  agg = collections.OrderedDict()
  for r in csv.DictReader(open(path)):
    a = agg.setdefault(short(r['Kernel_Name']), dict(n=0, d=[], vg=r.get('VGPR_Count') or r.get('Arch_VGPR_Count'),
                                                      av=r.get('Accum_VGPR_Count'), lds=r.get('LDS_Block_Size'),
                                                      scr=r.get('Scratch_Size')))
    a['n'] += 1
    if a['n'] > skip:
      a['d'].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
  tot = sum(sum(a['d']) for a in agg.values())
  print('| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | lds B | scratch |')
  print('|---|---|---|---|---|---|---|---|---|---|---|')
  for name, a in sorted(agg.items(), key=lambda kv: -sum(kv[1]['d'])):
    d = a['d']
    if not d:
      continue
    print('| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f | %s | %s | %s | %s |' % (
        name, len(d), sum(d) / 1e6, sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3, 100.0 * sum(d) / tot,
        a['vg'], a['av'], a['lds'], a['scr']))
  if '--by-grid' in sys.argv:
    sub = sys.argv[sys.argv.index('--by-grid') + 1]
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
      if sub in short(r['Kernel_Name']):
        key = (short(r['Kernel_Name']), r.get('Grid_Size') or r.get('Grid_Size_X'), r.get('Workgroup_Size') or r.get('Workgroup_Size_X'))
        rows.setdefault(key, []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    print('\n| kernel | grid | workgroup | calls | avg us | total ms |\n|---|---|---|---|---|---|')
    for (name, g, wg), d in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
      print('| `%s` | %s | %s | %d | %.1f | %.3f |' % (name, g, wg, len(d), sum(d) / len(d) / 1e3, sum(d) / 1e6))
  print('\ntotal kernel time %.3f ms over %d dispatches' % (tot / 1e6, sum(len(a['d']) for a in agg.values())))


if __name__ == '__main__':
  main()
