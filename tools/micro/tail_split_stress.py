"""Stress test of the tail split hand-over (igemm.hip): the workspace is poisoned with NaN before every
launch, so a slice summed before it was visible shows up as NaN / a wrong value."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from advoc_amd import conv
dev = torch.device('cuda')
torch.manual_seed(0)
B, h, w, cin, cout = 65, 32, 65, 32, 64
x = torch.randn(B, h, w, cin, device=dev)
wt = torch.randn(4, 4, cin, cout, device=dev) * 0.05
bias = torch.randn(cout, device=dev)
oh, ow = 16, 33
os.environ['ADVOC_IGEMM_TAIL'] = '0'
yp = torch.empty(B, oh, ow, cout, device=dev)
conv.Layer(conv.CONV, x, yp, wt, bias, stride=(2, 2), pad=(1, 1), in_act=conv.ACT_LRELU).forward()
os.environ['ADVOC_IGEMM_TAIL'] = '1'
y = torch.empty(B, oh, ow, cout, device=dev)
L = conv.Layer(conv.CONV, x, y, wt, bias, stride=(2, 2), pad=(1, 1), in_act=conv.ACT_LRELU)
print('ws bytes', L.struct.workspace_bytes)
ws = list(conv.Layer._workspaces.values())[0]
bad = 0
for it in range(400):
  ws.fill_(float('nan'))
  y.fill_(float('nan'))
  L.forward()
  torch.cuda.synchronize()
  n_nan = int(torch.isnan(y).sum())
  err = float((y - yp).abs().max()) if n_nan == 0 else -1
  if n_nan or err > 1e-4:
    bad += 1
    if bad < 6: print('iter', it, 'nan', n_nan, 'err', err)
print('bad', bad, 'of 400')
