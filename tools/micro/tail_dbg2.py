import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch
import test_hip_conv as T
from advoc_amd import conv
dev = torch.device('cuda')
for case in T.TAIL:
  c = T.build_case(case)
  x0 = c['x0'].to(dev); x1 = c['x1'].to(dev) if c['x1'] is not None else None
  w, b, dy = c['w'].to(dev), c['b'].to(dev), c['dy'].to(dev)
  mask = c['mask'].to(dev) if c['mask'] is not None else None
  cout = w.shape[3] if c['kind'] == 0 else w.shape[2]
  for use_b in (None, b):
    y = torch.full((x0.shape[0], c['oh'], c['out_w'], cout), float('nan'), device=dev)
    L = conv.Layer(c['kind'], x0, y, w, use_b, x1=x1, in_w=c['in_w'], stride=c['stride'], pad=c['pad'],
                   in_act=c['act'], drop_mask=mask, drop_scale=2. if mask is not None else 0.)
    L.forward(); torch.cuda.synchronize()
    nan = torch.isnan(y)
    print(case[0], 'bias' if use_b is not None else 'nobias', 'ws', L.struct.workspace_bytes, 'nan', int(nan.sum()),
          'of', y.numel(), L.kernel_name(0))
    if nan.any():
      idx = nan.nonzero()
      print('  first', idx[0].tolist(), 'last', idx[-1].tolist(), 'images', sorted(set(idx[:, 0].tolist()))[:10],
            'chans', sorted(set(idx[:, 3].tolist()))[:5], len(set(idx[:, 3].tolist())))
