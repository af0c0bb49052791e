"""The reference known answers that CANNOT be reproduced in this build's containers, kept visible as skipped tests
("List B"): each test computes what the oracle gives with the stand-ins that are available, and skips with
the reason and the distance -- so that the numbers stay on the test report instead of sitting unread in
tests/golden/known_answers.json.  Everything these constants depend on beyond the repo's own arithmetic is a third-party
resampler or TensorFlow itself:

  * reference tests/test_spectral.py:20-25 loads its 22.05 kHz / 24 kHz fixtures through advoc.audioio.decode_audio with
    fs=..., i.e. librosa.core.resample (resampy 'kaiser_best'), which is not installed here (no network); the scipy
    polyphase resampler standing in for it gives features that agree to ~4e-5 in the mean but not to the 2-8 decimals of
    the constants;
  * test_spectral.py:137-153 runs the TensorFlow graph on those resampled inputs.

The constants that depend only on files the reference ships at their native rate (the 16 kHz SC09 clip, the r9y9 pickle)
ARE asserted: tests/test_oracle_spectral.py, tests/test_hip_spectral.py."""
import json
import os

import numpy as np
import pytest
from scipy.io import wavfile
from scipy.signal import resample_poly

from oracle import spectral_np as S


@pytest.fixture(scope='module')
def known(golden_dir):
  with open(os.path.join(golden_dir, 'known_answers.json')) as f:
    return json.load(f)


@pytest.fixture(scope='module')
def mono(golden_dir):
  fs, x = wavfile.read(os.path.join(golden_dir, 'mono.wav'))
  assert fs == 44100
  x = x.astype(np.float32) / 32768. if x.dtype == np.int16 else x.astype(np.float32)
  return x.reshape(x.shape[0], -1)[:, :1]


def _resampled(mono, fs_out):
  """scipy polyphase stand-in for librosa.core.resample(..., res_type='kaiser_best') (advoc/audioio.py:44-52)."""
  g = np.gcd(44100, fs_out)
  y = resample_poly(mono[:, 0].astype(np.float64), fs_out // g, 44100 // g).astype(np.float32)
  return y[:, None, None]


def test_stft_tf_row_of_the_resampled_fixture(known, mono):
  """test_spectral.py:73: sum |STFT| of the first 16 384 samples of the 22.05 kHz fixture = 160.60 (2 decimals)."""
  want = known['stft_tf_sc09']['unreproducible_row0_sum']
  x = _resampled(mono, 22050)[np.newaxis, :16384]
  got = float(np.abs(S.stft_tf(x, 1024, 256, pad_end=True)).sum(dtype=np.float64))
  if round(abs(got - want), 2) == 0:
    return      # a resampler that reproduces librosa's: the pin holds
  pytest.skip('reference constant %.2f needs librosa/resampy resampling of the 44.1 kHz fixture (absent); the scipy '
              'polyphase stand-in gives %.2f (%.2e relative)' % (want, got, abs(got - want) / want))


def test_tacotron2_melspec_constants(known, mono):
  """test_spectral.py:79-86: Tacotron-2 mel features of the 24 kHz fixture: sum 131.469, rows 200 / 40 (3 decimals)."""
  ka = known['tacotron2_unreproducible']
  x = _resampled(mono, 24000)
  mel = S.waveform_to_tacotron2_melspec(x)
  assert mel.dtype == np.float64
  got = (float(mel.sum()), float(mel[200].sum()), float(mel[40].sum()))
  want = (ka['sum'], ka['row200'], ka['row40'])
  if list(mel.shape) == ka['shape'] and all(round(abs(g - w), 3) == 0 for g, w in zip(got, want)):
    return
  pytest.skip('reference constants %s (shape %s) need librosa/resampy resampling to 24 kHz (absent); the scipy stand-in gives '
              '%s, shape %s' % (want, ka['shape'], tuple(round(g, 3) for g in got), list(mel.shape)))


def test_r9y9_melspec_eight_decimal_sums(known, mono, golden_dir):
  """test_spectral.py:89-106: sum of the r9y9 mel features of the 22.05 kHz fixture equals the sum of the pickle from the
  r9y9 code base to 8 decimals (the pickle's own sum IS asserted elsewhere; the features computed here are not comparable
  at that precision without the reference's resampler)."""
  want = known['r9y9_mono22']['pkl_sum_skip3']
  mel = S.waveform_to_r9y9_melspec(_resampled(mono, 22050))
  got = float(mel.sum())
  if round(abs(got - want), 8) == 0:
    return
  pytest.skip('reference: sum(melspec) == %.9f to 8 decimals; needs the librosa/resampy-resampled fixture (absent): the scipy '
              'stand-in gives %.6f (%.1e relative, %d frames)' % (want, got, abs(got - want) / abs(want), mel.shape[0]))


def test_r9y9_tf_graph_sums_and_error(known):
  """test_spectral.py:137-153: four float32 channel sums of the TensorFlow graph on the resampled fixtures and its summed
  absolute error against the r9y9 pickle, 0.00731311 (8 decimals)."""
  ka = known['r9y9_mono22']
  pytest.skip('reference constants %s and %.8f are outputs of the TensorFlow 1.12 graph on librosa-resampled inputs: neither '
              'TensorFlow nor the resampler is installed in the build or the GPU image'
              % (ka['tf_sums_unreproducible'], ka['tf_abs_err_unreproducible']))


def test_lws_and_envelope_constants():
  """test_spectral.py:156-175 (envelope L1 after LWS / GL10: 0.01737 / 0.01686) and :178-208 (waveform L1 after LWS:
  0.0004236908353): need lws==1.2 (third-party C++, absent) and the resampled fixture.  The Griffin-Lim constants of the
  same test ARE reproduced to 0.1 % (tests/test_hip_inversion.py); the LWS restatement here reaches 5.7e-4 where lws 1.2
  reaches 4.24e-4 on the complex-input call (NOTEBOOK.md section 2)."""
  pytest.skip('lws 1.2 (the reference default phase estimator) is a third-party C++ package that is not installed: LWS parity is '
              'unpinned; see NOTEBOOK.md section 2')
