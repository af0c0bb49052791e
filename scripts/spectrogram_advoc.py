#!/usr/bin/env python
"""Vocodes a directory of mel spectrograms (.npy [T, 80, 1] float64) with an AdVoc generator on
MI355X.  Same flags as the reference script (scripts/spectrogram_advoc.py:26-45) plus
--model_type; --meta_fp is accepted and ignored (there is no TF meta graph).

Output: <name>.npy with the generated magnitude spectrogram [T, 513, 1] float32 in --out_dir.
The reference goes on to estimate phase with LWS and write <name>.wav (:95-97); waveform
synthesis is the next row of this build."""
import glob
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

if __name__ == '__main__':
  from argparse import ArgumentParser
  import numpy as np
  from advoc_amd.infer import load_generator, vocode_melspec

  parser = ArgumentParser()
  parser.add_argument('--spec_dir', type=str, required=True, help='Directory of spectrograms')
  parser.add_argument('--out_dir', type=str, required=True, help='Directory for outputs')
  parser.add_argument('--model_ckpt', type=str, help='Adversarial vocoder checkpoint')
  parser.add_argument('--meta_fp', type=str, help='(ignored) TF meta graph filepath')
  parser.add_argument('--fs', type=int, help='Sample rate')
  parser.add_argument('--subseq_len', type=int, help='model subseq length')
  parser.add_argument('--model_type', type=str, choices=['regular', 'small'])
  parser.set_defaults(spec_dir=None, out_dir=None, model_ckpt=None, meta_fp=None, fs=22050,
                      subseq_len=256, model_type='regular')
  args = parser.parse_args()

  if not os.path.isdir(args.out_dir):
    os.makedirs(args.out_dir)
  if args.model_ckpt is None:
    raise NotImplementedError('the pseudo-inverse + LWS heuristic (reference :48-50,77-78) needs LWS '
                              'phase reconstruction, which is not built yet; pass --model_ckpt')
  model = load_generator(args.model_ckpt, args.model_type, args.subseq_len, args.fs)
  for spec_fp in sorted(glob.glob(os.path.join(args.spec_dir, '*.npy'))):
    name = os.path.splitext(os.path.split(spec_fp)[1])[0]
    gen_mag = vocode_melspec(model, np.load(spec_fp))
    np.save(os.path.join(args.out_dir, name + '.npy'), gen_mag)
