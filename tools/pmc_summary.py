#!/usr/bin/env python
"""Per-kernel HBM-side traffic from two rocprofv3 counter passes (one --pmc FETCH_SIZE, one --pmc
WRITE_SIZE, each with --kernel-trace --output-format csv; never combined with other trace domains).

    python tools/pmc_summary.py FETCH_DIR/x_counter_collection.csv WRITE_DIR/y_counter_collection.csv \
        [--json profiles/r01_traffic.json] > profiles/r01_x_pmc_hbm.md

Units / corrections (MI355X_MICROARCH.md, "HBM"): both counters are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of 16-byte-per-lane streaming reads, so it is doubled; WRITE_SIZE is taken
as is (checked here on l1_kernel: 2 x 16.8 MB read -> FETCH_SIZE 16.4 MB, 16.8 MB written ->
WRITE_SIZE 16.5 MB; weight-gradient kernel reading 195 MB of 128-byte rows once -> 97 MB raw).
Calibration caveat for the gather GEMM: its A loads are 64-byte segments (4 lanes x 16 B per pixel row);
a pointwise instance that reads 270 MB exactly once reports 168 MB raw (factor 1.6, not 2), so the
doubled figure over-states that kernel's traffic by ~25 %.  Infinity-Cache hits are counted: this is
L2-miss traffic, an upper bound on HBM bytes."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from prof_summary import short


def collect(path):
  d = collections.defaultdict(list)
  for r in csv.DictReader(open(path)):
    d[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
  return d


def main():
  fetch, write = collect(sys.argv[1]), collect(sys.argv[2])
  out = {}
  print('| kernel | launches | FETCH_SIZE avg MiB (raw) | WRITE_SIZE avg MiB | traffic avg MB = 2*fetch + write |')
  print('|---|---|---|---|---|')
  for k in sorted(fetch, key=lambda k: -sum(fetch[k])):
    f = sum(fetch[k]) / len(fetch[k])
    w = sum(write[k]) / len(write[k]) if k in write else 0.
    traffic = (2 * f + w) * 1024
    out[k] = dict(launches=len(fetch[k]), fetch_kib_raw=f, write_kib=w, traffic_bytes=traffic)
    print('| `%s` | %d | %.2f | %.2f | %.1f |' % (k, len(fetch[k]), f / 1024, w / 1024, traffic / 1e6))
  if '--json' in sys.argv:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out['_workload'] = dict(model=os.environ.get('PMC_MODEL', 'regular'), batch=int(os.environ.get('PMC_BATCH', '64')),
                            kernel_source_sha16=bench.kernel_source_sha16(),     # bench.py withholds `traffic` when the sources moved on
                            command='python bench.py --steps 3 --warmup 1 --no-cpu-baseline --train-only --prof-steps 0')
    json.dump(out, open(sys.argv[sys.argv.index('--json') + 1], 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
  main()
