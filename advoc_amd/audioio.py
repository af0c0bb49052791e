"""WAV decode / encode on the host (drop-in for the reference's advoc.audioio).

Mirrors /root/reference/advoc/audioio.py:9-68 (decode_audio) and :71-93
(save_as_wav): same signatures, shapes ([nsamps, 1, nch] float32), error
types.  Only the ``fastwav=True`` branch is implemented -- both shipped data
configs set ``fastwav,1`` (datacfg/ljspeech.txt:3, datacfg/sc09.txt:3); the
librosa/resample branch (audioio.py:37-47) is out of scope
and raises.
"""
import numpy as np
from scipy.io import wavfile as _wavfile

_PCM16_SCALE = 32768.


def _read_standard_wav(fp):
  try:
    return _wavfile.read(fp)
  except Exception:
    raise ValueError('Error encountered when decoding WAV file.')


def decode_audio(fp, fs=None, mono=False, normalize=False, fastwav=False):
  """Decodes an audio file path into a float32 array of shape [nsamps, 1, nch].

  Returns (fs, x).  Raises ValueError for undecodable files, for a sample-rate
  mismatch (the fast path cannot resample) and for non PCM16 / float32 WAVs.
  """
  if not fastwav:
    raise NotImplementedError(
        'advoc_amd.audioio only implements the fastwav=True decode path '
        '(librosa/resampy are not part of the MI355X hot path).')

  file_fs, samples = _read_standard_wav(fp)
  if fs is not None and fs != file_fs:
    raise ValueError('Fastwav cannot resample audio.')

  if samples.dtype == np.int16:
    samples = samples.astype(np.float32)
    samples /= _PCM16_SCALE
  elif samples.dtype != np.float32:
    raise ValueError('Fastwav cannot process atypical WAV files.')

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
