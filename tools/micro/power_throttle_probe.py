#!/usr/bin/env python
"""Is the matrix pipe's 0.75 occupancy of patch_gemm_h3_kernel<1,0> a property of the schedule or of the power limit?
The SAME launch (D layer_4 forward, 128 images) on operands of decreasing switching activity, probe build
(ADVOC_HIP_LIB=.../libadvoc_hip_clk.so: workgroup life in shader cycles and the clock, printed by the kernel):
    random normal | random with the low planes zero (values rounded to fp16: h1 = 0) | constant 1.0 | all zero.
If the cycle count falls with the activity while the instruction stream is the same, the lost quarter is throttling
(cycles in which the pipe is not allowed to start an MFMA), not rendezvous.   python tools/micro/power_throttle_probe.py CASE(0..3)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from h3_sweep_shapes import build

L, dy, dx0, dx1 = build('d4b')
x, w = L.x0, L.weight
x_rand, w_rand = x.clone(), w.clone()
cases = [('random normal', lambda: (x.copy_(x_rand), w.copy_(w_rand))),
         ('fp16-exact values (h1 planes zero)', lambda: (x.copy_(x_rand.half().float()), w.copy_(w_rand.half().float()))),
         ('constant 1.0 / 0.05', lambda: (x.fill_(1.0), w.fill_(0.05))),
         ('all zero', lambda: (x.zero_(), w.zero_()))]
which = int(sys.argv[1]) if len(sys.argv) > 1 else 0      # one case per process: the kernel's printf and Python's do not interleave
name, prep = cases[which]
prep()
torch.cuda.synchronize()
print('CASE %s' % name, flush=True)
for _ in range(8):
  L.forward()
torch.cuda.synchronize()
