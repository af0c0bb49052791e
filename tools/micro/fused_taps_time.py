#!/usr/bin/env python
"""The <= 2-column layers of AdVoc-full at B = 64 (generator decoder_1 forward, discriminator layer_5 forward at 2B and B,
layer_1 backward-data): fused_taps_kernel against the two launches it replaces, and the LDS budget of its patches.
One process per setting (the library reads its switches once)."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
import torch, bench
from advoc_amd import conv
dev = torch.device('cuda')
def layer(kind, B, H, W, c0, c1, cout, stride, pad, act, trim, oh, ow):
  x0 = torch.randn(B, H, W + trim, c0, device=dev)
  x1 = torch.randn(B, H, W, c1, device=dev) if c1 else None
  w = torch.randn(4, 4, c0 + c1, cout, device=dev) * 0.05 if kind == 0 else torch.randn(4, 4, cout, c0 + c1, device=dev) * 0.05
  y = torch.empty(B, oh, ow, cout, device=dev)
  L = conv.Layer(kind, x0, y, w, torch.zeros(cout, device=dev), x1=x1, in_w=W, stride=stride, pad=pad, in_act=act)
  return L, x0, x1, y
res = {}
L, x0, x1, y = layer(1, 64, 128, 257, 64, 64, 1, (2, 2), (1, 1), 2, 1, 256, 513)
res['decoder_1 fwd'] = (L.kernel_name(0), bench.event_timed(torch, L.forward, 20), (x0[:, :, :257].numel() + x1.numel() + y.numel()) * 4)
del L, x0, x1, y
for B in (128, 64):
  L, x0, x1, y = layer(0, B, 31, 63, 512, 0, 1, (1, 1), (1, 1), 1, 0, 30, 62)
  res['layer_5 fwd B=%%d' %% B] = (L.kernel_name(0), bench.event_timed(torch, L.forward, 20), (x0.numel() + y.numel()) * 4)
  del L, x0, x1, y
L, x0, x1, y = layer(0, 64, 256, 513, 1, 1, 64, (2, 2), (1, 1), 0, 0, 128, 256)
dy = torch.randn_like(y); dx0 = torch.zeros_like(x0); dx1 = torch.zeros_like(x1)
res['layer_1 bwdD'] = (L.kernel_name(1), bench.event_timed(torch, lambda: L.backward_data(dy, dx0, dx1), 20), (dy.numel() + dx0.numel() + dx1.numel()) * 4)
res['layer_1 bwdD G step'] = (L.kernel_name(1), bench.event_timed(torch, lambda: L.backward_data(dy, None, dx1, accum1=True), 20), (dy.numel() + 2 * dx1.numel()) * 4)
print(json.dumps(res))
''' % ROOT
settings = [a for a in sys.argv[1:]] or ['ADVOC_FUSED_TAPS=0', '', 'ADVOC_FUSED_TAPS_LDS_KB=52', 'ADVOC_FUSED_TAPS_LDS_KB=39', 'ADVOC_FUSED_TAPS_LDS_KB=150']
for st in settings:
  env = dict(os.environ)
  for kv in st.split():
    k, v = kv.split('=')
    env[k] = v
  r = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True)
  if r.returncode:
    print(st, r.stderr[-600:])
    continue
  d = json.loads(r.stdout.strip().splitlines()[-1])
  print('%-32s' % (st or '(default)'), '  '.join('%s %s %.0f us %.2f TB/s' % (k, v[0][:22], v[1] * 1e3, v[2] / v[1] / 1e9) for k, v in d.items()), flush=True)
