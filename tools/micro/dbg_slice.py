import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from h3_sweep_shapes import build
from advoc_amd import _lib
L, dy, dx0, dx1 = build('d4b')
print('name', L.kernel_name(0), L.kernel_name(1))
L2, *_ = build('dec2m')
print('name', L2.kernel_name(0), L2.kernel_name(1))
try:
  L.forward(); torch.cuda.synchronize(); print('fwd ok')
except Exception as e:
  print('ERR', e)
print(_lib.load().advoc_last_hip_error() if hasattr(_lib.load(), 'advoc_last_hip_error') else 'no last error api')
