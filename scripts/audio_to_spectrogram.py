#!/usr/bin/env python
"""Feature dump: every audio file of a directory becomes one dB-normalised mel spectrogram
(`<stem>.npy`, float64 [frames, 80, 1], r9y9 preset) computed by the HIP extractor on the GPU.

Command line of the reference's scripts/audio_to_spectrogram.py (flags :15-29): --wave_dir, --out_dir,
--fs (default 22050), --data_fast_wav.  Only the fast WAV decoder exists here (advoc_amd.audioio); a file
that needs resampling or a non-WAV codec makes decode_audio raise, as documented there."""
import argparse
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))


def parse_args(argv=None):
  ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
  ap.add_argument('--wave_dir', required=True, help='directory with the audio files')
  ap.add_argument('--out_dir', required=True, help='directory the .npy spectrograms are written to')
  ap.add_argument('--fs', type=int, default=22050, help='sample rate the features are defined at')
  ap.add_argument('--data_fast_wav', action='store_true', help='decode standard WAV files with scipy (fast path)')
  return ap.parse_args(argv)


def dump_features(wave_dir, out_dir, fs, fast_wav):
  import numpy as np
  from advoc_amd.audioio import decode_audio
  from advoc_amd.spectral import waveform_to_r9y9_melspec
  target = Path(out_dir)
  target.mkdir(parents=True, exist_ok=True)
  written = 0
  for src in sorted(Path(wave_dir).iterdir()):
    if not src.is_file():
      continue
    _, samples = decode_audio(str(src), fs=fs, mono=True, normalize=True, fastwav=fast_wav)
    np.save(str(target / (src.stem + '.npy')), waveform_to_r9y9_melspec(samples, fs=fs))
    written += 1
  return written


if __name__ == '__main__':
  a = parse_args()
  dump_features(a.wave_dir, a.out_dir, a.fs, a.data_fast_wav)
