#!/usr/bin/env python
"""Matrix-pipe / issue counters per kernel from a rocprofv3 --pmc pass (SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, --output-format csv).
    python tools/sq_summary.py DIR/*_counter_collection.csv > profiles/rNN_x_pmc_mfma.md
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): fraction of the matrix pipes' cycles that
carried an MFMA at the clock the kernel actually ran at (GRBM_GUI_ACTIVE is summed over the 8 XCDs; the SQ_WAIT_* /
SQ_ACTIVE_* / SQ_WAVE_CYCLES counters are in quad-cycles, MI355X_MICROARCH.md).
(r5) Checked against the instruction count of patch_gemm_h3_kernel<1,0> and the shader clock probed inside the step
(profiles/r05_winograd_and_layer4_account.md): 0.729 by this column, 0.75 by instruction count x 32 cycles / probed cycles."""
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
    if 'GRBM_GUI_ACTIVE' not in m or m['GRBM_GUI_ACTIVE'] <= 0:
      continue
    cyc = m['GRBM_GUI_ACTIVE'] / 8.0
    wc = max(m.get('SQ_WAVE_CYCLES', 0.0), 1.0)
    rows.append((cyc * n, k, n, cyc, m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (cyc * 1024.0),
                 m.get('SQ_WAIT_ANY', 0.0) / wc, m.get('SQ_WAIT_INST_ANY', 0.0) / wc, m.get('SQ_ACTIVE_INST_ANY', 0.0) / wc,
                 m.get('SQ_WAVES', 0.0)))
  print('| kernel | launches | avg cycles | MFMA busy | waves parked (WAIT_ANY) | issue stall (WAIT_INST_ANY) | issuing (ACTIVE_INST_ANY) | waves |')
  print('|---|---|---|---|---|---|---|---|')
  for _, k, n, cyc, busy, wa, wi, ac, waves in sorted(rows, reverse=True):
    print('| `%s` | %d | %.3g | %.3f | %.3f | %.3f | %.3f | %d |' % (k, n, cyc, busy, wa, wi, ac, waves))


if __name__ == '__main__':
  main()
