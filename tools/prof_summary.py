#!/usr/bin/env python
"""Summarises a rocprofv3 --kernel-trace result database (rocpd sqlite) as a per-kernel table:
calls, total / average / min / max duration, share.  Usage:
    python tools/prof_summary.py gpurun_out/prof_x/x_results.db [--skip-first N] > profiles/x.md
--skip-first drops the first N dispatches of every kernel (warm-up iterations)."""
import re
import sqlite3
import sys


def short(name):
  name = re.sub(r'advoc::\(anonymous namespace\)::', '', name)
  name = re.sub(r'\(anonymous namespace\)::', '', name)
  name = re.sub(r'^void ', '', name)
  name = re.sub(r'\(.*\)$', '', name)
  return name[:110]


def main():
  db = sys.argv[1]
  skip = int(sys.argv[sys.argv.index('--skip-first') + 1]) if '--skip-first' in sys.argv else 0
  cur = sqlite3.connect(db).cursor()
  rows = cur.execute('select name, start, end, vgpr_count, accum_vgpr_count, lds_size, scratch_size, '
                     'grid_x, grid_y, grid_z, workgroup_x from kernels order by start').fetchall()
  agg = {}
  for name, s, e, vg, av, lds, scr, gx, gy, gz, wx in rows:
    a = agg.setdefault(name, dict(n=0, d=[], vg=vg, av=av, lds=lds, scr=scr))
    a['n'] += 1
    if a['n'] > skip:
      a['d'].append(e - s)
  tot = sum(sum(a['d']) for a in agg.values())
  print('| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | lds B | scratch |')
  print('|---|---|---|---|---|---|---|---|---|---|---|')
  for name, a in sorted(agg.items(), key=lambda kv: -sum(kv[1]['d'])):
    d = a['d']
    if not d:
      continue
    print('| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f | %s | %s | %s | %s |' % (
        short(name), len(d), sum(d) / 1e6, sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3,
        100.0 * sum(d) / tot, a['vg'], a['av'], a['lds'], a['scr']))
  print('\ntotal kernel time %.3f ms over %d dispatches' % (tot / 1e6, sum(len(a['d']) for a in agg.values())))


if __name__ == '__main__':
  main()
