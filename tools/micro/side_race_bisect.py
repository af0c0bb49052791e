#!/usr/bin/env python
"""Bisects the side-stream / batch-norm discrepancy: device-wide syncs after selected weight-gradient calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from advoc_amd import _lib, conv
from advoc_amd.model import AdvocSmall, Modes
dev = torch.device('cuda')
g = torch.Generator().manual_seed(9)
target = torch.rand(16, 128, 513, 1, generator=g) * 2
x = target * (0.5 + torch.rand(16, 128, 513, 1, generator=g)) - 0.1
x, target = x.to(dev), target.to(dev)
orig_bw = conv.Layer.backward_weight
orig_bd = conv.Layer.backward_data
SYNC = set()
SYNC_BD = set()


def bw(self, *a, **k):
  r = orig_bw(self, *a, **k)
  if id(self) in SYNC:
    torch.cuda.synchronize()
  return r


def bd(self, *a, **k):
  if id(self) in SYNC_BD:
    torch.cuda.synchronize()
  return orig_bd(self, *a, **k)


conv.Layer.backward_weight = bw
conv.Layer.backward_data = bd


def run(side, which=None, env=None):
  os.environ['ADVOC_WGRAD_STREAM'] = '1' if side else '0'
  for k, v in (env or {}).items():
    os.environ[k] = v
  _lib.reload_env()
  m = AdvocSmall(Modes.TRAIN)
  m.subseq_len = 128
  m.train_batch_size = 16
  m.use_batchnorm = True
  m.build(batch_size=16, seed=4)
  st = m._built
  SYNC.clear(); SYNC_BD.clear()
  if which:
    kind, idxs = which
    for name in ('d_layers_real', 'd_layers_fake'):
      if kind in ('all', name, name + ':bd'):
        for i in idxs:
          (SYNC_BD if kind.endswith(':bd') else SYNC).add(id(st[name][i]))
  cap = {}
  orig = m._bn_backward
  def bnb(b, g, **kw):
    tag = 'c%d_%s' % (b['c'], 'fake' if b['z'].data_ptr() != st['d_act'][[64,128,256].index(b['c']) + 1].data_ptr() else 'real')
    cap['pre_' + tag] = g.clone()
    r = orig(b, g, **kw)
    cap['work_' + tag] = b['work'].clone()
    return r
  if os.environ.get('CAP'):
    m._bn_backward = bnb
  m((x, target))
  m.d_step((x, target))
  torch.cuda.synchronize()
  out = {k: v.detach().clone() for k, v in st['d_G'].items()}
  for i in range(5):
    out['g_act%d_real' % i] = st['g_d_act'][i][:16].clone()
    out['g_act%d_fake' % i] = st['g_d_act'][i][16:].clone()
  out.update(cap)
  for k in (env or {}):
    os.environ.pop(k)
  _lib.reload_env()
  return out


def rel(a, b):
  return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


ref = run(0)
KEY = ['discriminator/layer_3/batch_normalization/gamma', 'discriminator/layer_2/batch_normalization/gamma',
       'discriminator/layer_1/conv2d/kernel', 'discriminator/layer_3/conv2d/kernel']
print('layers:', [(i, [l.kernel_name(d) for d in range(3)]) for i, l in enumerate(AdvocSmall(Modes.TRAIN).__class__ and [])])


def report(tag, which=None, env=None, n=3):
  worst = {k: 0.0 for k in KEY}
  for _ in range(n):
    r = run(1, which, env)
    for k in KEY:
      worst[k] = max(worst[k], rel(r[k], ref[k]))
  print('%-44s %s' % (tag, '  '.join('%.1e' % worst[k] for k in KEY)), flush=True)


r = run(0)
print('%-44s %s' % ('serial rerun', '  '.join('%.1e' % rel(r[k], ref[k]) for k in KEY)))
for t in range(24):
  r = run(1)
  print('trial', t)
  for k in sorted(r):
    if k.startswith('g_act') or k.startswith('pre_') or k.startswith('work_') or 'gamma' in k:
      d = (r[k].double() - ref[k].double())
      print('   %-50s rel %.2e  max|d| %.2e  nonzero diffs %d of %d' % (k, rel(r[k], ref[k]), float(d.abs().max()), int((d != 0).sum()), d.numel()))
      if k == 'pre_c128_fake' and int((d != 0).sum()):
        big = (d.abs() > 1e-3 * ref[k].abs().max().double()).nonzero()
        nz = (d != 0).nonzero()
        for nm, ix in (('nonzero', nz), ('big', big)):
          if len(ix):
            print('      %s: n %s h %s w %s c [%d..%d] count %d' % (nm, sorted(set(ix[:, 0].tolist())), sorted(set(ix[:, 1].tolist())), sorted(set(ix[:, 2].tolist()))[:70], int(ix[:, 3].min()), int(ix[:, 3].max()), len(ix)))
        if len(big):
          i0 = tuple(big[0].tolist())
          print('      first big: idx', i0, 'side', float(r[k][i0]), 'serial', float(ref[k][i0]))
          n0, h0, w0 = i0[:3]
          print('      side row :', [round(float(v), 9) for v in r[k][n0, h0, w0, :8]])
          print('      ref  row :', [round(float(v), 9) for v in ref[k][n0, h0, w0, :8]])
