#!/usr/bin/env python
"""Round-4 race hunt, part 2: AdVoc-small with batch norm, D step with the weight gradients on the side stream against the
one-stream step; with captures of every tensor of the fake pass in program order (which is the first to differ).  (The r4
session also split the culprit -- thin_wgrad_kernel's bias sums: memset / sums in the kernel / reduce, LDS atomics vs plain
stores vs global atomics -- with temporary switches in thin.hip; results in NOTEBOOK.md section 5.)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from advoc_amd import _lib, conv
from advoc_amd.model import Advoc, AdvocSmall, Modes

TRIALS = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B, T = 16, 128
dev = torch.device('cuda')
gen = torch.Generator().manual_seed(9)
target = torch.rand(B, T, 513, 1, generator=gen) * 2
x = target * (0.5 + torch.rand(B, T, 513, 1, generator=gen)) - 0.1
x, target = x.to(dev), target.to(dev)


def run(side, env=None, capture=False):
  os.environ['ADVOC_WGRAD_STREAM'] = '1' if side else '0'
  os.environ['ADVOC_WGRAD_H3_ORDERED'] = '2'
  for k, v in (env or {}).items():
    os.environ[k] = v
  _lib.reload_env()
  m = AdvocSmall(Modes.TRAIN)
  m.subseq_len, m.train_batch_size, m.use_batchnorm = T, B, True
  m.build(batch_size=B, seed=4)
  st = m._built
  cap = {}
  if capture:
    for li in (4, 3, 2, 1):
      lay = st['d_layers_fake'][li]
      def wrap(lay=lay, li=li, orig=lay.backward_data):
        def bd(dy, *a, **k):
          cap['L%d.dy_in' % li] = dy.clone()
          r = orig(dy, *a, **k)
          if len(lay._img) >= 4:
            cap['L%d.dy_img' % li] = lay._img[2].clone()
            cap['L%d.dy_hdr' % li] = lay._img[3].clone()
          cap['L%d.dx' % li] = a[0].clone()
          return r
        return bd
      lay.backward_data = wrap()
    orig_bn = m._bn_backward
    def bnb(b, g, **kw):
      fake = any(b is v for v in st['d_bn_fake'].values())
      tag = 'bn_c%d_%s' % (b['c'], 'fake' if fake else 'real')
      cap[tag + '.scale'] = b['scale'].clone()
      cap[tag + '.shift'] = b['shift'].clone()
      r = orig_bn(b, g, **kw)
      cap[tag + '.g_out'] = g.clone()
      cap[tag + '.work'] = b['work'].clone()
      return r
    m._bn_backward = bnb
  m((x, target))
  m.d_step((x, target))
  torch.cuda.synchronize()
  out = {'g_d_act%d' % i: st['g_d_act'][i].clone() for i in range(5)}
  for k, v in st['d_G'].items():
    out['dG:' + k] = v.detach().clone()
  out.update(cap)
  for k in (env or {}):
    os.environ.pop(k)
  return out


def ndiff(a, b):
  return int((a != b).sum())


ref = run(False)
print('serial rerun equal:', all(torch.equal(v, r2) for (k, v), r2 in zip(ref.items(), run(False).values()) if 'bias' not in k and 'layer_5' not in k))
bad = 0
for t in range(TRIALS):
  r = run(True)
  if not torch.equal(r['g_d_act2'], ref['g_d_act2']):
    bad += 1
print('%d / %d side-stream trials with a wrong g_d_act2' % (bad, TRIALS), flush=True)
if os.environ.get('NO_CAPTURE'):
  sys.exit(0)
print('--- captures (all parts on) ---')
refc = run(False, capture=True)
shown = 0
for t in range(3 * TRIALS):
  r = run(True, capture=True)
  if torch.equal(r['g_d_act2'], refc['g_d_act2']):
    continue
  shown += 1
  print('trial', t, 'differs; tensors in program order:')
  order = ['L4.dy_in', 'L4.dx', 'bn_c256_fake.scale', 'bn_c256_fake.g_out', 'L3.dy_in', 'L3.dy_img', 'L3.dy_hdr', 'bn_c128_fake.scale',
           'bn_c128_fake.shift', 'L3.dx', 'bn_c128_fake.g_out', 'bn_c128_fake.work', 'L2.dy_in', 'L2.dx']
  for k in order:
    if k in r:
      a, b = r[k], refc[k]
      d = (a.double() - b.double()) if a.dtype.is_floating_point else (a.long() - b.long()).double()
      nz = (d != 0).nonzero()
      line = '   %-22s wrong %8d of %9d  max|d| %.3e  (max |ref| %.3e)' % (k, len(nz), d.numel(), float(d.abs().max()), float(b.double().abs().max()))
      if len(nz) and a.dim() == 4:
        line += '  n %s h %s w %s c [%d..%d]' % (sorted(set(nz[:, 0].tolist()))[:6], sorted(set(nz[:, 1].tolist()))[:20],
                                                 sorted(set(nz[:, 2].tolist()))[:40], int(nz[:, 3].min()), int(nz[:, 3].max()))
      elif len(nz):
        line += '  idx %s' % nz[:, 0].tolist()[:24]
      print(line)
  if shown >= 3:
    break
