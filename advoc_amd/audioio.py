"""WAV decode / encode on the host (drop-in for the reference's advoc.audioio).

Mirrors /root/reference/advoc/audioio.py:9-68 (decode_audio) and :71-93
(save_as_wav): same signatures, shapes ([nsamps, 1, nch] float32), error
types.  The ``fastwav=True`` branch is bit-exact against the reference (both
shipped data configs set ``fastwav,1``: datacfg/ljspeech.txt, datacfg/sc09.txt).
The ``fastwav=False`` branch of the reference goes through librosa
(audioio.py:37-47), which is not in this image: here it decodes WAV files of any
PCM width and resamples with scipy's polyphase filter, so it returns the same
shapes and the same samples when no resampling is needed, but resampled values
differ from librosa's kaiser filter (parity unpinned for that case); compressed
formats raise ValueError.
"""
from fractions import Fraction

import numpy as np
from scipy.io import wavfile as _wavfile
from scipy.signal import resample_poly as _resample_poly

_PCM16_SCALE = 32768.


def _read_standard_wav(fp):
  try:
    return _wavfile.read(fp)
  except Exception:
    raise ValueError('Error encountered when decoding WAV file.')


def _decode_general(fp, fs):
  """The fastwav=False branch: any-width PCM WAV, optional resample to ``fs``."""
  try:
    file_fs, samples = _wavfile.read(fp)
  except Exception:
    raise ValueError('Error encountered when decoding audio file.')
  if samples.dtype == np.int16:
    samples = samples.astype(np.float32) / np.float32(_PCM16_SCALE)
  elif samples.dtype == np.int32:
    samples = (samples.astype(np.float64) / 2147483648.).astype(np.float32)
  elif samples.dtype == np.uint8:
    samples = (samples.astype(np.float32) - 128.) / np.float32(128.)
  else:
    samples = samples.astype(np.float32)
  if fs is not None and fs != file_fs:
    ratio = Fraction(int(fs), int(file_fs))
    samples = _resample_poly(samples.astype(np.float64), ratio.numerator, ratio.denominator,
                             axis=0).astype(np.float32)
    file_fs = fs
  return file_fs, samples


def decode_audio(fp, fs=None, mono=False, normalize=False, fastwav=False):
  """Decodes an audio file path into a float32 array of shape [nsamps, 1, nch].

  Returns (fs, x).  Raises ValueError for undecodable files, for a sample-rate
  mismatch (the fast path cannot resample) and for non PCM16 / float32 WAVs.
  """
  if fastwav:
    file_fs, samples = _read_standard_wav(fp)
    if fs is not None and fs != file_fs:
      raise ValueError('Fastwav cannot resample audio.')
    if samples.dtype == np.int16:
      samples = samples.astype(np.float32)
      samples /= _PCM16_SCALE
    elif samples.dtype != np.float32:
      raise ValueError('Fastwav cannot process atypical WAV files.')
  else:
    file_fs, samples = _decode_general(fp, fs)

  nch = 1 if samples.ndim == 1 else samples.shape[1]
  x = samples.reshape(samples.shape[0], 1, nch)

  if mono:
    x = x.mean(axis=2, keepdims=True)

  if normalize:
    peak = np.abs(x).max() if x.size else 0.
    if peak > 0:
      x = x / peak

  return file_fs, np.ascontiguousarray(x, dtype=np.float32)


def save_as_wav(fp, fs, x):
  """Writes a float32 [nsamps, 1, 1] waveform as signed 16-bit PCM."""
  if np.ndim(x) != 3:
    raise ValueError('Incorrect number of input dimesions.')
  _, nfeats, nch = x.shape
  if nfeats != 1:
    raise ValueError('Incorrect input dimesions.')
  if nch != 1:
    raise NotImplementedError('Can only save monaural WAV for now.')

  pcm = np.clip(x[:, 0, 0].astype(np.float32) * np.float32(_PCM16_SCALE), -32768., 32767.)
  _wavfile.write(fp, fs, pcm.astype(np.int16))
