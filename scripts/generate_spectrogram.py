#!/usr/bin/env python
"""Samples mel spectrograms from a MelspecGAN generator on MI355X.  Same flags as the reference
script (scripts/generate_spectrogram.py:12-29): --out_dir --ckpt_fp --meta_fp (ignored: there is
no TF meta graph) --n --b; plus --dim and --seed.  Writes <index zero-padded to 9 digits>.npy,
float32 [64, 80, 1] in [0, 1] (the G_z tensor, :49-56), the input format of
scripts/spectrogram_advoc.py --subseq_len 64.

--ckpt_fp is a TensorFlow checkpoint prefix (model.ckpt-N, read by advoc_amd/tf_checkpoint.py) or a
torch file holding {TF variable name: tensor}; without it the generator has its random
initialisation (useful only for throughput checks)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

if __name__ == '__main__':
  from argparse import ArgumentParser
  import numpy as np
  import torch
  from advoc_amd.melspecgan import MelspecGANGenerator
  from advoc_amd import tf_checkpoint

  parser = ArgumentParser()
  parser.add_argument('--out_dir', type=str, required=True)
  parser.add_argument('--ckpt_fp', type=str)
  parser.add_argument('--meta_fp', type=str, help='(ignored) TF meta graph filepath')
  parser.add_argument('--n', type=int)
  parser.add_argument('--b', type=int)
  parser.add_argument('--dim', type=int)
  parser.add_argument('--seed', type=int)
  parser.set_defaults(ckpt_fp=None, meta_fp=None, n=1000, b=100, dim=64, seed=None)
  args = parser.parse_args()

  if not os.path.isdir(args.out_dir):
    os.makedirs(args.out_dir)
  G = MelspecGANGenerator(dim=args.dim)
  if args.ckpt_fp is not None:
    if tf_checkpoint.is_tf_checkpoint(args.ckpt_fp):
      print('Restored from step {}'.format(G.load_tf_checkpoint(args.ckpt_fp)))
    else:
      G.load_state_dict(torch.load(args.ckpt_fp, map_location='cpu'))
  gen = torch.Generator().manual_seed(args.seed) if args.seed is not None else None
  for i in range(0, args.n, args.b):
    b = min(args.b, args.n - i)
    z = torch.randn(b, 100, generator=gen)
    G_z = G(z, denorm=True).cpu().numpy()
    for j, s in enumerate(G_z):
      np.save(os.path.join(args.out_dir, '{}.npy'.format(str(j + i).zfill(9))), s.astype(np.float32))
