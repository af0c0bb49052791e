#!/usr/bin/env python
"""Fixed cost of a gather-GEMM launch: time vs contraction depth at a fixed grid (1024 workgroups)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from grid_quant import bench

for B in (31, 62):
  for cin in (16, 32, 64, 128, 256, 512):
    us, tf, name, M = bench(B, 32, 65, cin, 256, reps=40)
    print('B=%d cin=%4d  K tiles=%4d  %8.1f us  %6.1f TFLOP/s  %s' % (B, cin, cin, us, tf, name))
