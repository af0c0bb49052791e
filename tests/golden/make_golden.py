"""Regenerates the reference-derived golden vectors under tests/golden/.

Run in the BUILD container only (needs /root/reference, read-only):
    python tests/golden/make_golden.py

What it does
  1. imports the reference's advoc/audioio.py (the only reference module that
     imports without TF/lws/librosa) and records decode_audio(..., fastwav=True)
     outputs for the three WAV fixtures -> audioio_golden.npz
  2. records the reference tests' known-answer constants for the spectral path
     (tests/test_spectral.py:33-46,74-76,140; tests/test_audioio.py:20-25)
     -> known_answers.json
The WAV/MP3 files and mono_22k_r9y9.npy (the pickled float64 [80,325] array of
tests/audio/mono_22k_r9y9.pkl re-saved as .npy) are data files the reference's
own tests hold; they are copied verbatim.
"""
import importlib.util
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def main():
  spec = importlib.util.spec_from_file_location('ref_audioio', os.path.join(REF, 'advoc', 'audioio.py'))
  ref = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(ref)

  out = {}
  for name in ['sc09', 'mono', 'stereo']:
    fp = os.path.join(HERE, name + '.wav')
    for tag, kw in [('raw', {}), ('mono', {'mono': True}), ('norm', {'mono': True, 'normalize': True})]:
      fs, x = ref.decode_audio(fp, fastwav=True, **kw)
      key = '{}_{}'.format(name, tag)
      out[key + '_fs'] = np.int64(fs)
      out[key + '_shape'] = np.array(x.shape, dtype=np.int64)
      out[key + '_min'] = np.float64(x.min())
      out[key + '_max'] = np.float64(x.max())
      out[key + '_sum'] = np.float64(x.astype(np.float64).sum())
      out[key + '_head'] = x[:64].copy()
      out[key + '_tail'] = x[-64:].copy()
  np.savez(os.path.join(HERE, 'audioio_golden.npz'), **out)

  known = {
      'source': 'reference tests/test_spectral.py and tests/test_audioio.py (constants only)',
      'stft_sc09': {'shape_pad': [63, 513, 1], 'shape_nopad': [60, 513, 1], 'shape_pad384': [64, 513, 1],
                    'sum': 2148.69, 'sum_row33': 55.45, 'sum_row40': 20.35, 'places': 2},
      'stft_tf_sc09': {'sum': 2148.69, 'sum_row33': 55.45, 'sum_row40': 20.35, 'places': 2,
                       'unreproducible_row0_sum': 160.60},
      'r9y9_mono22': {'shape': [322, 80, 1], 'pkl_sum_skip3': 5121.489431680473,
                      'tf_sums_unreproducible': [18328.508, 18332.746, 18319.934, 5121.489],
                      'tf_abs_err_unreproducible': 0.00731311},
      'tacotron2_unreproducible': {'shape': [300, 80, 1], 'sum': 131.469, 'row200': 0.644, 'row40': 0.0},
      'audioio_mono': {'fs': 44100, 'shape': [164864, 1, 1], 'min': -0.474823, 'max': 0.397278, 'places': 6},
      'audioio_stereo': {'shape': [164864, 1, 2]},
      'mono_22k_shape': [82432, 1, 1],
      'stft_nopad_mono22_shape': [319, 513, 1],
  }
  with open(os.path.join(HERE, 'known_answers.json'), 'w') as f:
    json.dump(known, f, indent=1, sort_keys=True)


if __name__ == '__main__':
  main()
