#!/usr/bin/env python
"""Vocodes a directory of mel spectrograms (.npy [T, 80, 1] float64) with an AdVoc generator on
MI355X.  Same flags as the reference script (scripts/spectrogram_advoc.py:26-45) plus
--model_type; --meta_fp is accepted and ignored (there is no TF meta graph).

Output: <name>.wav (PCM16, save_as_wav) in --out_dir, as the reference writes (:95-97), plus
<name>.npy with the generated magnitude spectrogram [T, 513, 1] float32 when --save_mag is given.
Phase estimation: LWS as in the reference (:95; the GPU restatement of the published algorithm, parity unpinned) by
default; --phase_estimation gl<N> runs Griffin-Lim (advoc/spectral.py:294-311) instead.  Without --model_ckpt the mel
pseudo-inverse heuristic (:48-50,77-78) is used."""
import glob
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

if __name__ == '__main__':
  from argparse import ArgumentParser
  import numpy as np
  from advoc_amd.audioio import save_as_wav
  from advoc_amd.infer import load_generator, vocode_melspec
  from advoc.audioio import save_as_wav  # noqa: F811  (the reference's import line, scripts/spectrogram_advoc.py:10)
  from advoc.spectral import magspec_to_waveform_griffin_lim, magspec_to_waveform_lws, r9y9_melspec_to_waveform

  parser = ArgumentParser()
  parser.add_argument('--spec_dir', type=str, required=True, help='Directory of spectrograms')
  parser.add_argument('--out_dir', type=str, required=True, help='Directory for outputs')
  parser.add_argument('--model_ckpt', type=str, help='Adversarial vocoder checkpoint')
  parser.add_argument('--meta_fp', type=str, help='(ignored) TF meta graph filepath')
  parser.add_argument('--fs', type=int, help='Sample rate')
  parser.add_argument('--subseq_len', type=int, help='model subseq length')
  parser.add_argument('--model_type', type=str, choices=['regular', 'small'])
  parser.add_argument('--phase_estimation', type=str, help='lws (default) or gl<N>: Griffin-Lim with N iterations')
  parser.add_argument('--save_mag', action='store_true', help='also write the generated magnitudes (.npy)')
  parser.set_defaults(spec_dir=None, out_dir=None, model_ckpt=None, meta_fp=None, fs=22050,
                      subseq_len=256, model_type='regular', phase_estimation='lws')
  args = parser.parse_args()

  if not os.path.isdir(args.out_dir):
    os.makedirs(args.out_dir)
  if args.phase_estimation != 'lws' and args.phase_estimation[:2] != 'gl':
    raise ValueError('--phase_estimation: lws or gl<N>')
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
      if args.phase_estimation == 'lws':
        wave = magspec_to_waveform_lws(gen_mag.astype('float64'), 1024, 256)
      else:
        wave = magspec_to_waveform_griffin_lim(gen_mag.astype('float64'), 1024, 256, int(args.phase_estimation[2:]))
    save_as_wav(os.path.join(args.out_dir, name + '.wav'), args.fs, wave)
