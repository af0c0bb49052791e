#!/usr/bin/env python
"""Error of the three matrix paths against float64 on whole layers (forward and backward-data):
  fp32 MFMA chain | register-split bf16 triples, six products | fp16 pair images, three products.
    python tools/micro/h3_numerics.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from advoc_amd import _lib, conv
dev = torch.device('cuda')


def setenv(**kw):
  for k, v in kw.items():
    if v is None:
      os.environ.pop(k, None)
    else:
      os.environ[k] = str(v)
  _lib.reload_env()


def rel(a, b):
  return float((a.double().cpu() - b).norm() / b.norm())


CASES = [('conv s2 128->256 (encoder_4-like)', 0, (8, 32, 65), 128, 0, 256, (2, 2), 1),
         ('conv s1 128->256 (layer_4-like)', 0, (8, 32, 64), 128, 0, 256, (1, 1), 1),
         ('deconv 256+256->128 (decoder_4-like)', 1, (8, 16, 33), 256, 256, 128, (2, 2), 2)]
for name, kind, (B, H, W), c0, c1, cout, stride, act in CASES:
  g = torch.Generator().manual_seed(0)
  # activations with a wide dynamic range (log-normal envelope) and small-magnitude gradients
  x0 = torch.randn(B, H, W + (1 if c1 else 0), c0, generator=g) * torch.exp(1.5 * torch.randn(B, H, W + (1 if c1 else 0), c0, generator=g))
  x1 = torch.randn(B, H, W, c1, generator=g) if c1 else None
  w = torch.randn(4, 4, *( (c0 + c1, cout) if kind == 0 else (cout, c0 + c1)), generator=g) * 0.02
  if kind == 0:
    oh, ow = (H - 1, W - 1) if stride == (1, 1) else (-(-H // 2), -(-W // 2))
  else:
    oh, ow = 2 * H, 2 * W
  dy = torch.randn(B, oh, ow, cout, generator=g) * 1e-6
  # float64 reference on the CPU through torch
  xin = torch.cat([x0[:, :, :W]] + ([x1] if c1 else []), dim=3).double()
  a = torch.nn.functional.leaky_relu(xin, 0.2) if act == 1 else torch.relu(xin)
  a = a.requires_grad_(True)
  if kind == 0:
    pad = (1, 1, 1, 1) if stride == (1, 1) else (1, 2 if W % 2 else 1, 1, 2 if H % 2 else 1)
    ap = torch.nn.functional.pad(a.permute(0, 3, 1, 2), pad)
    y64 = torch.nn.functional.conv2d(ap, w.double().permute(3, 2, 0, 1), stride=stride).permute(0, 2, 3, 1)[:, :oh, :ow]
  else:
    y64 = torch.nn.functional.conv_transpose2d(a.permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), stride=2, padding=1).permute(0, 2, 3, 1)
  (ga,) = torch.autograd.grad(y64, a, dy.double())
  row = []
  for tag, env in (('fp32 MFMA', dict(ADVOC_IGEMM_X6=0)), ('bf16x3 (6)', dict(ADVOC_H3=0)), ('fp16x2 (3)', dict(ADVOC_H3_MIN_TILES=1))):
    setenv(ADVOC_IGEMM_X6=None, ADVOC_H3=None, ADVOC_H3_MIN_TILES=None)
    setenv(**env)
    y = torch.empty(B, oh, ow, cout, device=dev)
    L = conv.Layer(kind, x0.to(dev), y, w.to(dev), None, x1=x1.to(dev) if c1 else None, in_w=W, stride=stride, pad=(1, 1), in_act=act)
    L.forward()
    dx0 = torch.zeros_like(x0, device=dev)
    dx1 = torch.zeros_like(x1, device=dev) if c1 else None
    # backward-data WITHOUT the activation gate: compare d/d(act(x))
    L.struct.in_act = 0
    L.backward_data(dy.to(dev), dx0, dx1)
    gx = torch.cat([dx0[:, :, :W]] + ([dx1] if c1 else []), dim=3)
    row.append('%s fwd %.2e bwd %.2e (%s)' % (tag, rel(y, y64.detach()), rel(gx, ga), L.kernel_name(0).split('<')[0]))
  print('%-40s %s' % (name, ' | '.join(row)), flush=True)
