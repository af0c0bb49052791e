#!/usr/bin/env python
"""Round-4 hunt for the side-stream corruption (NOTEBOOK.md section 5): d_step / g_step with the weight gradients on the
side stream against serial execution, EVERY gradient tensor and every backward-data output compared element by element,
under a list of variants that each remove one suspect.  Prints, per variant, how many trials differed and where the
wrong elements of the first differing tensors sit.

  python tools/micro/side_race_r4.py [small|full] [trials]
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from advoc_amd import _lib, conv
from advoc_amd.model import Advoc, AdvocSmall, Modes

which = sys.argv[1] if len(sys.argv) > 1 else 'small'
TRIALS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B, T = 16, 128
dev = torch.device('cuda')
gen = torch.Generator().manual_seed(9)
target = torch.rand(B, T, 513, 1, generator=gen) * 2
x = target * (0.5 + torch.rand(B, T, 513, 1, generator=gen)) - 0.1
x, target = x.to(dev), target.to(dev)

NOISE = {'on': False}
_noise_src = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()
_noise_dst = torch.empty_like(_noise_src)


def run(side, bn, env=None, noise=False, steps=('d', 'g')):
  os.environ['ADVOC_WGRAD_STREAM'] = '1' if side else '0'
  os.environ['ADVOC_WGRAD_H3_ORDERED'] = '2'
  for k, v in (env or {}).items():
    os.environ[k] = v
  _lib.reload_env()
  m = (AdvocSmall if which == 'small' else Advoc)(Modes.TRAIN)
  m.subseq_len, m.train_batch_size, m.use_batchnorm = T, B, bn
  m.build(batch_size=B, seed=4)
  st = m._built
  m((x, target))
  out = {}
  nstream = torch.cuda.Stream(device=dev) if noise else None
  for step in steps:
    if noise:      # an HBM-bound stranger on another stream while the step runs
      nstream.wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(nstream):
        for _ in range(40):
          _noise_dst.copy_(_noise_src)
    (m.d_step if step == 'd' else m.g_step)((x, target))
    torch.cuda.synchronize()
    if step == 'd':
      for k, v in st['d_G'].items():
        out['dG:' + k] = v.detach().clone()
      for i in range(5):
        out['d:g_d_act%d' % i] = st['g_d_act'][i].clone()
    else:
      for k, v in st['g_G'].items():
        out['gG:' + k] = v.detach().clone()
      for i in range(5):
        out['g:g_d_act%d' % i] = st['g_d_act'][i][B:].clone()
      for i, t in enumerate(st['g_enc']):
        out['g:g_enc%d' % i] = t.clone()
      for i, t in st['g_dec'].items():
        out['g:g_dec%d' % i] = t.clone()
  for k in (env or {}):
    os.environ.pop(k)
  _lib.reload_env()
  return out


def describe(k, a, b):
  d = (a.double() - b.double())
  nz = (d != 0).nonzero()
  scale = float(b.abs().max())
  msg = '    %-58s wrong %d of %d  max|d| %.2e (tensor max %.2e)' % (k, len(nz), d.numel(), float(d.abs().max()), scale)
  if a.dim() == 4 and len(nz):
    msg += '\n       n %s  h [%d..%d] (%d distinct)  w %s  c [%d..%d]' % (
        sorted(set(nz[:, 0].tolist()))[:16], int(nz[:, 1].min()), int(nz[:, 1].max()), len(set(nz[:, 1].tolist())),
        sorted(set(nz[:, 2].tolist()))[:24], int(nz[:, 3].min()), int(nz[:, 3].max()))
  return msg


def campaign(tag, bn, env=None, noise=False, side=True, ref_env=None):
  ref = run(False, bn, ref_env if ref_env is not None else env)
  ref2 = run(False, bn, ref_env if ref_env is not None else env)
  unstable = [k for k in ref if not torch.equal(ref[k], ref2[k])]
  bad_trials = 0
  first = None
  keys_bad = {}
  for t in range(TRIALS):
    r = run(side, bn, env, noise)
    bad = [k for k in ref if k not in unstable and not torch.equal(r[k], ref[k])]
    if bad:
      bad_trials += 1
      for k in bad:
        keys_bad[k] = keys_bad.get(k, 0) + 1
      if first is None:
        first = [describe(k, r[k], ref[k]) for k in bad[:6]]
  print('%-60s bn %d: %d / %d trials differ%s' % (tag, bn, bad_trials, TRIALS,
                                                   ('  (serial rerun unstable: %s)' % unstable[:4]) if unstable else ''), flush=True)
  if first:
    print('  tensors that differed (trials): %s' % sorted(keys_bad.items(), key=lambda kv: -kv[1])[:10])
    for line in first:
      print(line)
  return bad_trials


names = [(i, [l.kernel_name(d) for d in range(3)]) for i, l in enumerate([])]
NOJOIN = {}      # (r3 joined the side stream between the two batch-norm passes; r4 removed the join with the cause)
print('model', which, 'B', B, 'T', T, 'trials', TRIALS, flush=True)
# 0. what ships: BN off; BN on with the inter-pass join
campaign('default (no BN)', False)
campaign('BN, inter-pass join (what ships)', True)
# 1. the r3 reproduction: BN, no join
n0 = campaign('BN, NO inter-pass join', True, NOJOIN)
# 2. a stranger instead of the weight gradients: everything on one stream + an HBM-bound copy loop on another
campaign('serial + HBM noise on another stream', True, None, noise=True, side=False)
campaign('serial + HBM noise, no BN', False, None, noise=True, side=False)
# 3. suspects removed one at a time (all: BN, no join)
for tag, env in (
    ('remainder K slices meet in atomics (REM_WS=0)', {'ADVOC_H3_REM_WS': '0'}),
    ('no remainder launch (PATCH_REM=0)', {'ADVOC_H3_PATCH_REM': '0'}),
    ('no K split anywhere (IGEMM_SPLITK=0)', {'ADVOC_IGEMM_SPLITK': '0'}),
    ('no patch kernels (H3_PATCH=0)', {'ADVOC_H3_PATCH': '0'}),
    ('no persistent patch workgroups', {'ADVOC_H3_PATCH_PERSIST': '0'}),
    ('thin wgrad bias off', {'ADVOC_THIN_WGRAD_BIAS': '0'}),
    ('no image weight gradient (WGRAD_H3=0)', {'ADVOC_WGRAD_H3': '0'}),
    ('no delayed scale / emit', {'ADVOC_DELAYED_SCALE': '0', 'ADVOC_EMIT_IMAGES': '0'}),
):
  e = dict(NOJOIN)
  e.update(env)
  campaign('BN no join, ' + tag, True, e)
