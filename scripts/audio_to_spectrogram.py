#!/usr/bin/env python
"""Directory of WAVs -> directory of r9y9 mel spectrograms (.npy [T, 80, 1] float64), features
computed by the HIP extractor.  Same flags as the reference script
(scripts/audio_to_spectrogram.py:15-29); only --data_fast_wav decoding is available."""
import glob
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

if __name__ == '__main__':
  from argparse import ArgumentParser
  import numpy as np
  from advoc_amd.audioio import decode_audio
  from advoc_amd.spectral import waveform_to_r9y9_melspec

  parser = ArgumentParser()
  parser.add_argument('--wave_dir', type=str, required=True, help='Directory of audio files')
  parser.add_argument('--out_dir', type=str, required=True, help='Directory for spectrograms')
  parser.add_argument('--fs', type=int, help='Sample rate')
  parser.add_argument('--data_fast_wav', action='store_true', dest='data_fast_wav',
                      help='If set, provides faster loading of standard WAV files via scipy')
  parser.set_defaults(wave_dir=None, out_dir=None, fs=22050, data_fast_wav=False)
  args = parser.parse_args()

  if not os.path.isdir(args.out_dir):
    os.makedirs(args.out_dir)
  for wave_fp in sorted(glob.glob(os.path.join(args.wave_dir, '*'))):
    name = os.path.splitext(os.path.split(wave_fp)[1])[0]
    _, wave = decode_audio(wave_fp, fs=args.fs, fastwav=args.data_fast_wav, mono=True, normalize=True)
    np.save(os.path.join(args.out_dir, name + '.npy'), waveform_to_r9y9_melspec(wave, fs=args.fs))
