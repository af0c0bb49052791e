"""Pins oracle/spectral_np.py against the reference's own known-answer constants
(reference tests/test_spectral.py:27-46, 49-76, 89-106) -- CPU only."""
import json
import os

import numpy as np
import pytest
from scipy.io import wavfile

from oracle import spectral_np as S


@pytest.fixture(scope='module')
def known(golden_dir):
  with open(os.path.join(golden_dir, 'known_answers.json')) as f:
    return json.load(f)


@pytest.fixture(scope='module')
def sc09(golden_dir):
  fs, x = wavfile.read(os.path.join(golden_dir, 'sc09.wav'))
  assert fs == 16000 and x.dtype == np.int16
  return (x.astype(np.float32) / 32768.)[:, None, None]


def test_window_is_lws_half_sample_hann():
  w = S.lws_hann_default(1024, 256, np.float64)
  k = np.arange(1024)
  np.testing.assert_allclose(w ** 2, (0.5 - 0.5 * np.cos(2 * np.pi * (k + 0.5) / 1024)) * 0.5, rtol=1e-14)
  # symmetric, and sqrt-Hann at 75 % overlap is a tight frame: sum of w^2 over the 4 shifts == 1
  np.testing.assert_allclose(w, w[::-1], rtol=1e-13)
  np.testing.assert_allclose(sum(np.roll(w ** 2, 256 * i) for i in range(4)), 1.0, rtol=1e-13)


def test_stft_numpy_known_answers(sc09, known):
  ka = known['stft_sc09']
  X = S.stft(sc09, 1024, 256, pad_end=True)
  assert X.dtype == np.complex128 and list(X.shape) == ka['shape_pad']
  assert list(S.stft(sc09, 1024, 256, pad_end=False).shape) == ka['shape_nopad']
  xp = np.pad(sc09, [[0, 384], [0, 0], [0, 0]], 'constant')
  X = S.stft(xp, 1024, 256, pad_end=True)
  assert list(X.shape) == ka['shape_pad384']
  mag = np.abs(X)
  assert mag.dtype == np.float64
  assert round(abs(mag.sum() - ka['sum']), ka['places']) == 0
  assert round(abs(mag[33].sum() - ka['sum_row33']), ka['places']) == 0
  assert round(abs(mag[40].sum() - ka['sum_row40']), ka['places']) == 0


def test_stft_tf_known_answers(sc09, known):
  ka = known['stft_tf_sc09']
  x = np.pad(sc09[np.newaxis], [[0, 0], [0, 384], [0, 0], [0, 0]], 'constant')
  X = S.stft_tf(x, 1024, 256, pad_end=True)
  assert X.dtype == np.complex64 and X.shape == (1, 64, 513, 1)
  mag = np.abs(X)
  assert mag.dtype == np.float32
  assert round(abs(float(mag[0].sum(dtype=np.float64)) - ka['sum']), 2) == 0
  assert round(abs(float(mag[0, 33].sum()) - ka['sum_row33']), 2) == 0
  assert round(abs(float(mag[0, 40].sum()) - ka['sum_row40']), 2) == 0
  assert S.stft_tf(sc09[np.newaxis], 1024, 256).shape == (1, 63, 513, 1)
  # the f32 graph is within 1e-6 relative L2 of exact arithmetic on the same f32 inputs
  exact = S.stft_mag_f64(x, 1024, 256)
  assert np.linalg.norm(mag - exact) / np.linalg.norm(exact) < 1e-6


def test_wrong_hann_conventions_fail_the_pin(sc09):
  """The constants discriminate the window convention: periodic / symmetric
  Hann variants miss at 2 decimals, so the pin is meaningful."""
  xp = np.pad(sc09[:, 0, 0], [0, 384 + 768]).astype(np.float64)
  idx = np.arange(64)[:, None] * 256 + np.arange(1024)[None, :]
  k = np.arange(1024)
  good = np.sqrt((0.5 - 0.5 * np.cos(2 * np.pi * (k + 0.5) / 1024)) * 0.5)
  periodic = np.sqrt((0.5 - 0.5 * np.cos(2 * np.pi * k / 1024)) * 0.5)
  symmetric = np.sqrt((0.5 - 0.5 * np.cos(2 * np.pi * k / 1023)) * 0.5)
  sums = [np.abs(np.fft.rfft(xp[idx] * w, axis=1)).sum() for w in (good, periodic, symmetric)]
  assert round(abs(sums[0] - 2148.69), 2) == 0
  assert round(abs(sums[1] - 2148.69), 2) != 0
  assert round(abs(sums[2] - 2148.69), 2) != 0


def test_frame_counts(known):
  assert S.num_frames_tf(16000, 1024, 256, True) == 63
  assert S.num_frames_tf(16000, 1024, 256, False) == 59
  assert S.num_frames_lws(16000, 1024, 256) == 60
  assert S.num_frames_lws(82432, 1024, 256) == known['stft_nopad_mono22_shape'][0]
  assert S.num_frames_tf(0, 1024, 256, True) == 0
  assert S.num_frames_tf(100, 1024, 256, False) == 0


def test_mel_filterbank_structure():
  W = S.create_mel_filterbank(22050, 1024, fmin=125, fmax=7600, n_mels=80)
  assert W.shape == (80, 513) and W.dtype == np.float64
  bm = S.mel_bin_map(W)
  assert tuple(bm[0]) == (6, 9) and tuple(bm[79]) == (329, 352)
  assert (W >= 0).all()
  # Slaney area normalisation: each triangle integrates to ~1 over Hz
  df = 22050 / 2 / 512
  area = W.sum(axis=1) * df
  assert np.all(np.abs(area - 1) < 0.2)
  # bands are contiguous runs of non-zeros and ordered
  for m in range(80):
    nz = np.nonzero(W[m])[0]
    assert nz[0] == bm[m, 0] and nz[-1] == bm[m, 1] and len(nz) == nz[-1] - nz[0] + 1
  assert np.all(np.diff(bm[:, 0]) >= 0) and np.all(np.diff(bm[:, 1]) >= 0)
  Wi = S.create_inverse_mel_filterbank(22050, 1024, fmin=125, fmax=7600, n_mels=80)
  assert Wi.shape == (513, 80)
  np.testing.assert_allclose(W @ Wi, np.eye(80), atol=1e-9)


def test_r9y9_fixture_loose(golden_dir, known):
  """mono_22k_r9y9 pins the mel + dB-norm restatement semantically.  The reference resamples
  mono.wav 44.1k->22.05k with librosa/resampy (absent here); with scipy's polyphase resampler
  as a stand-in the features agree to ~4e-5 mean (SURVEY.md §8c list A)."""
  from scipy.signal import resample_poly
  g = np.load(os.path.join(golden_dir, 'mono_22k_r9y9.npy'))
  assert g.shape == (80, 325) and g.dtype == np.float64
  ref = np.swapaxes(g, 0, 1)[3:, :, np.newaxis]
  assert abs(ref.sum() - known['r9y9_mono22']['pkl_sum_skip3']) < 1e-8
  fs, m = wavfile.read(os.path.join(golden_dir, 'mono.wav'))
  m22 = resample_poly(m.astype(np.float64) / 32768., 1, 2).astype(np.float32)[:, None, None]
  assert list(m22.shape) == known['mono_22k_shape']
  mel = S.waveform_to_r9y9_melspec(m22)
  assert mel.dtype == np.float64 and list(mel.shape) == known['r9y9_mono22']['shape']
  assert np.abs(mel - ref).mean() < 1e-4 and np.abs(mel - ref).max() < 5e-4
  mel32 = S.waveform_to_r9y9_melspec_tf(m22[np.newaxis])
  assert mel32.dtype == np.float32 and mel32.shape == (1, 322, 80, 1)
  assert np.abs(mel32[0].astype(np.float64) - mel).max() < 1e-5


def test_errors():
  with pytest.raises(ValueError):
    S.stft(np.zeros((10, 2, 1), np.float32), 1024, 256)
  with pytest.raises(NotImplementedError):
    S.stft(np.zeros((10, 1, 2), np.float32), 1024, 256)
  with pytest.raises(ValueError):
    S.waveform_to_melspec(np.zeros((10, 1, 1), np.float64), 22050, 1024, 256)
  with pytest.raises(NotImplementedError):
    S.waveform_to_melspec_tf(np.zeros((1, 2048, 1, 1), np.float32), 22050, 1024, 256,
                             norm_allow_clipping=False)
