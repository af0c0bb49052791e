#!/usr/bin/env python
"""One layer shape, forward (and optionally backward-data) a few times under the current ADVOC_* environment:
the target of rocprofv3 counter passes.   python tools/micro/h3_one.py SHAPE [reps] [dirs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from h3_sweep_shapes import build
name = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dirs = sys.argv[3] if len(sys.argv) > 3 else 'f'
L, dy, dx0, dx1 = build(name)
for _ in range(reps):
  if 'f' in dirs:
    L.forward()
  if 'd' in dirs:
    L.backward_data(dy, dx0, dx1)
torch.cuda.synchronize()
print(L.kernel_name(0), L.kernel_name(1))
