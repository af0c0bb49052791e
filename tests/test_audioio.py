"""advoc_amd.audioio against vectors produced by the reference's own advoc/audioio.py
(tests/golden/make_golden.py) and the reference's test constants
(reference tests/test_audioio.py:19-40, 80-96).  CPU only (host code)."""
import os
import tempfile

import numpy as np
import pytest

from advoc_amd.audioio import decode_audio, save_as_wav


@pytest.fixture(scope='module')
def gold(golden_dir):
  return np.load(os.path.join(golden_dir, 'audioio_golden.npz'))


@pytest.mark.parametrize('name', ['sc09', 'mono', 'stereo'])
@pytest.mark.parametrize('tag,kw', [('raw', {}), ('mono', {'mono': True}),
                                    ('norm', {'mono': True, 'normalize': True})])
def test_decode_matches_reference_bit_exact(golden_dir, gold, name, tag, kw):
  fs, x = decode_audio(os.path.join(golden_dir, name + '.wav'), fastwav=True, **kw)
  k = '{}_{}'.format(name, tag)
  assert x.dtype == np.float32
  assert fs == int(gold[k + '_fs'])
  assert tuple(x.shape) == tuple(gold[k + '_shape'])
  assert float(x.min()) == float(gold[k + '_min']) and float(x.max()) == float(gold[k + '_max'])
  assert np.array_equal(x[:64], gold[k + '_head']) and np.array_equal(x[-64:], gold[k + '_tail'])
  assert abs(x.astype(np.float64).sum() - float(gold[k + '_sum'])) < 1e-9


def test_reference_constants(golden_dir):
  wav = os.path.join(golden_dir, 'mono.wav')
  fs, x = decode_audio(wav, fastwav=True)
  assert fs == 44100 and x.shape == (164864, 1, 1)
  assert round(abs(float(x.min()) + 0.474823), 6) == 0
  assert round(abs(float(x.max()) - 0.397278), 6) == 0
  with pytest.raises(ValueError):
    decode_audio(wav, fs=22050, fastwav=True)
  _, xn = decode_audio(wav, normalize=True, fastwav=True)
  assert round(abs(float(np.abs(xn).max()) - 1.), 8) == 0
  _, xs = decode_audio(os.path.join(golden_dir, 'stereo.wav'), fastwav=True)
  assert xs.shape == (164864, 1, 2)
  _, xm = decode_audio(os.path.join(golden_dir, 'stereo.wav'), mono=True, fastwav=True)
  assert xm.shape == (164864, 1, 1)
  with pytest.raises(ValueError):
    decode_audio(os.path.join(golden_dir, 'mono.mp3'), fastwav=True)


def test_save_as_wav_roundtrip(golden_dir):
  fs, x = decode_audio(os.path.join(golden_dir, 'mono.wav'), fastwav=True)
  with tempfile.NamedTemporaryFile(suffix='.wav') as tf:
    with pytest.raises(ValueError):
      save_as_wav(tf.name, fs, x[:, 0])
    with pytest.raises(ValueError):
      save_as_wav(tf.name, fs, np.concatenate([x, x], axis=1))
    with pytest.raises(NotImplementedError):
      save_as_wav(tf.name, fs, np.concatenate([x, x], axis=2))
    save_as_wav(tf.name, fs, x)
    fs2, x2 = decode_audio(tf.name, fastwav=True)
    assert fs2 == fs and np.array_equal(x, x2)
    # saturation instead of wrap-around
    save_as_wav(tf.name, fs, np.array([2., -2., 0.5], np.float32)[:, None, None])
    _, y = decode_audio(tf.name, fastwav=True)
    assert np.array_equal(y[:, 0, 0], np.array([32767, -32768, 16384], np.float32) / 32768.)


def test_empty_wave_roundtrip():
  with tempfile.NamedTemporaryFile(suffix='.wav') as tf:
    save_as_wav(tf.name, 16000, np.zeros((0, 1, 1), np.float32))
    fs, x = decode_audio(tf.name, fastwav=True, normalize=True)
    assert fs == 16000 and x.shape == (0, 1, 1)


def test_general_decode_branch(golden_dir):
  """reference tests/test_audioio.py:41-52, 68-74: the non-fastwav branch returns the
  same samples for a WAV, and the reference's shape when resampling."""
  wav = os.path.join(golden_dir, 'mono.wav')
  fs_g, x_g = decode_audio(wav, fastwav=False)
  fs_f, x_f = decode_audio(wav, fastwav=True)
  assert fs_g == fs_f and type(fs_g) == type(fs_f)
  assert np.array_equal(x_g, x_f)
  fs, x = decode_audio(wav, fs=22050)
  assert fs == 22050 and x.shape == (82432, 1, 1) and x.dtype == np.float32
  # decimation by two of a band-limited clip keeps its envelope
  assert abs(float(np.abs(x).max()) - float(np.abs(x_f).max())) < 0.05
  with pytest.raises(ValueError):
    decode_audio(os.path.join(golden_dir, 'mono.mp3'))
