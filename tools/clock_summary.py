#!/usr/bin/env python
"""Shader clock per kernel instance inside the real train step, from the probe build's log:
    bash tools/micro/build_variant.sh clk "-DADVOC_CLOCK_PROBE"
    ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_clk.so python bench.py --train-only --no-cpu-baseline --steps 4 --prof-steps 0 > log
    python tools/clock_summary.py log > profiles/rNN_x_shader_clock.md
Every probed workgroup brackets its whole life with s_memtime (shader cycles) and s_memrealtime (100 MHz): clock = ratio.
For the persistent kernels (one workgroup per CU for the whole launch) the workgroup's life is the launch's length; launches of one
instance are clustered by that length (layers and batch sizes differ)."""
import collections
import re
import sys

rows = collections.defaultdict(list)
pat = re.compile(r'clk (<[\d,]+>|wgrad_h3_256) wg\s+\d+(?: of \d+)?: (\d+) cycles in (\d+) ticks')
for line in open(sys.argv[1]):
  m = pat.match(line)
  if m:
    rows[m.group(1)].append((int(m.group(2)), int(m.group(3))))
print('| kernel instance | launch class (workgroup life, cycles) | probed workgroups | clock GHz (mean) | min | max | us (mean) |')
print('|---|---|---|---|---|---|---|')
for k, v in sorted(rows.items(), key=lambda kv: -sum(c for c, _ in kv[1])):
  name = 'patch_gemm_h3_kernel' + k if k[0] == '<' else k + '_kernel'
  classes = []                     # launches of one instance differ by layer / batch: cluster the lives within 8 %
  for c, t in sorted(v):
    if classes and c <= 1.08 * classes[-1][0][0]:
      classes[-1].append((c, t))
    else:
      classes.append([(c, t)])
  for cl in sorted(classes, key=lambda cl: -cl[0][0]):
    ghz = [0.1 * c / t for c, t in cl]
    print('| `%s` | %.3g | %d | %.3f | %.3f | %.3f | %.1f |' % (name, sum(c for c, _ in cl) / len(cl), len(cl), sum(ghz) / len(ghz),
                                                          min(ghz), max(ghz), sum(t for _, t in cl) / len(cl) / 100.0))
