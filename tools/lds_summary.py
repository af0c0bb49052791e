#!/usr/bin/env python
"""LDS counters per kernel from a rocprofv3 --pmc pass (SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS
SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE, --output-format csv).  conflict = BANK_CONFLICT / IDX_ACTIVE: share of the LDS-array
cycles that were replays; lds busy = IDX_ACTIVE / (cycles x 256 CUs).
    python tools/lds_summary.py DIR/*_counter_collection.csv"""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_summary import short


def main():
  agg = collections.defaultdict(lambda: collections.defaultdict(list))
  for r in csv.DictReader(open(sys.argv[1])):
    agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
  rows = []
  for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    n = len(next(iter(d.values())))
    cyc = m.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
    if cyc <= 0:
      continue
    idx = max(m.get('SQ_LDS_IDX_ACTIVE', 0.0), 1.0)
    rows.append((cyc * n, k, n, cyc, m.get('SQ_LDS_BANK_CONFLICT', 0.0) / idx, idx / (cyc * 256.0),
                 m.get('SQ_INSTS_LDS', 0.0), m.get('SQ_WAIT_INST_LDS', 0.0), m.get('SQ_ACTIVE_INST_LDS', 0.0)))
  print('| kernel | launches | avg cycles | conflict share of LDS cycles | LDS array busy | LDS instructions | WAIT_INST_LDS | ACTIVE_INST_LDS |')
  print('|---|---|---|---|---|---|---|---|')
  for _, k, n, cyc, conf, busy, insts, wl, al in sorted(rows, reverse=True):
    print('| `%s` | %d | %.3g | %.3f | %.3f | %.3g | %.3g | %.3g |' % (k, n, cyc, conf, busy, insts, wl, al))


if __name__ == '__main__':
  main()
