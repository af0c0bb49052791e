#!/usr/bin/env python
"""Vocodes a directory of mel spectrograms (.npy [T, 80, 1] float64) with an AdVoc generator on
MI355X.  Same flags as the reference script (scripts/spectrogram_advoc.py:26-45) plus
--model_type; --meta_fp is accepted and ignored (there is no TF meta graph).

Output: <name>.wav (PCM16, save_as_wav) in --out_dir, as the reference writes (:95-97), plus
<name>.npy with the generated magnitude spectrogram [T, 513, 1] float32 when --save_mag is given.
Phase estimation: the reference uses LWS (third-party lws 1.2, not restated here); this build
runs Griffin-Lim on the GPU, --phase_estimation gl<N> (default gl60, the reference's own
alternative, advoc/spectral.py:294-311).  Without --model_ckpt the mel pseudo-inverse heuristic
(:48-50,77-78) is used."""
import glob
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

if __name__ == '__main__':
  from argparse import ArgumentParser
  import numpy as np
  from advoc_amd.audioio import save_as_wav
  from advoc_amd.infer import load_generator, vocode_melspec
  from advoc_amd.spectral import magspec_to_waveform_griffin_lim, r9y9_melspec_to_waveform

  parser = ArgumentParser()
  parser.add_argument('--spec_dir', type=str, required=True, help='Directory of spectrograms')
  parser.add_argument('--out_dir', type=str, required=True, help='Directory for outputs')
  parser.add_argument('--model_ckpt', type=str, help='Adversarial vocoder checkpoint')
  parser.add_argument('--meta_fp', type=str, help='(ignored) TF meta graph filepath')
  parser.add_argument('--fs', type=int, help='Sample rate')
  parser.add_argument('--subseq_len', type=int, help='model subseq length')
  parser.add_argument('--model_type', type=str, choices=['regular', 'small'])
  parser.add_argument('--phase_estimation', type=str, help='gl<N>: Griffin-Lim with N iterations')
  parser.add_argument('--save_mag', action='store_true', help='also write the generated magnitudes (.npy)')
  parser.set_defaults(spec_dir=None, out_dir=None, model_ckpt=None, meta_fp=None, fs=22050,
                      subseq_len=256, model_type='regular', phase_estimation='gl60')
  args = parser.parse_args()

  if not os.path.isdir(args.out_dir):
    os.makedirs(args.out_dir)
  if args.phase_estimation[:2] != 'gl':
    raise NotImplementedError('only Griffin-Lim phase estimation (gl<N>) is built; LWS is third-party')
  ngl = int(args.phase_estimation[2:])
  model = None
  if args.model_ckpt is not None:
    model = load_generator(args.model_ckpt, args.model_type, args.subseq_len, args.fs)
  for spec_fp in sorted(glob.glob(os.path.join(args.spec_dir, '*.npy'))):
    name = os.path.splitext(os.path.split(spec_fp)[1])[0]
    spec = np.load(spec_fp)
    if model is None:
      # (float64 is what melspec_to_waveform insists on, spectral.py:359-360; generated MelspecGAN
      #  spectrograms are stored as float32)
      wave = r9y9_melspec_to_waveform(spec.astype(np.float64), fs=args.fs, phase_estimation=args.phase_estimation)
    else:
      gen_mag = vocode_melspec(model, spec)
      if args.save_mag:
        np.save(os.path.join(args.out_dir, name + '.npy'), gen_mag)
      wave = magspec_to_waveform_griffin_lim(gen_mag.astype('float64'), 1024, 256, ngl)
    save_as_wav(os.path.join(args.out_dir, name + '.wav'), args.fs, wave)
