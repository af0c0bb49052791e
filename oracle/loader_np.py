"""numpy restatement of the reference's batch pipeline (deterministic paths only).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: the reference has no loader
test.  Restates /root/reference/advoc/loader.py:76-91 (decode), :99-130 (feature branches),
:133-186 (_parallel_slice) and :199-203 (batch, drop_remainder) plus tf.contrib.signal.frame
(TF 1.13: ceil(N/hop) frames with pad_end, else 1 + (N - L)//hop; pad_value fills the tail).
Shuffling / random offsets use TF's unseeded RNG in the reference and are not restated.
"""
import numpy as np
from scipy.io import wavfile

from oracle import spectral_np as S


def decode(fp, fs, mono=True, normalize=False):
  file_fs, x = wavfile.read(fp)
  if fs is not None and fs != file_fs:
    raise ValueError('Fastwav cannot resample audio.')
  if x.dtype == np.int16:
    x = x.astype(np.float32) / np.float32(32768.)
  x = x.reshape(x.shape[0], 1, -1)
  if mono:
    x = np.mean(x, 2, keepdims=True)
  if normalize:
    f = np.max(np.abs(x))
    if f > 0:
      x = x / f
  return x.astype(np.float32)


def frame(x, length, hop, pad_end, pad_value=0.):
  """tf.contrib.signal.frame along axis 0, written as an explicit loop."""
  n = x.shape[0]
  if pad_end:
    nframes = -(-n // hop)
  else:
    nframes = max(0, 1 + (n - length) // hop)
  out = np.full((nframes, length) + x.shape[1:], pad_value, dtype=x.dtype)
  for i in range(nframes):
    lo = i * hop
    hi = min(lo + length, n)
    if hi > lo:
      out[i, :hi - lo] = x[lo:hi]
  return out


def parallel_slice(features, audio, slice_len, audio_fs, feature_fs, overlap, pad_end, first_only):
  if overlap < 0:
    raise ValueError('Slice overlap must be nonnegative')
  slice_hop = int(round(slice_len * (1. - overlap)))
  if slice_hop < 1:
    raise ValueError('Overlap ratio too high')
  r = float(audio_fs) / float(feature_fs)
  alen = int(round(slice_len * r) + 1e-4)
  ahop = int(round(slice_hop * r) + 1e-4)
  fs_ = frame(features, slice_len, slice_hop, pad_end)
  as_ = frame(audio, alen, ahop, pad_end)
  if first_only:
    fs_, as_ = fs_[:1], as_[:1]
  n = min(fs_.shape[0], as_.shape[0])     # Dataset.zip stops at the shorter stream
  return fs_[:n], as_[:n]


def batches(fps, batch_size, slice_len, audio_fs=22050, audio_normalize=False, extract_type=None,
            nfft=1024, nhop=256, first_only=False, overlap=0., pad_end=False):
  """Deterministic pipeline (no shuffle / repeat / random offset) -> list of (feats, audio)."""
  ex = []
  for fp in fps:
    wav = decode(fp, audio_fs, True, audio_normalize)
    if extract_type is None:
      feats, ffs = wav, audio_fs
    elif extract_type == 'magspec':
      feats, ffs = np.abs(S.stft_tf(wav[None], nfft, nhop))[0].astype(np.float32), audio_fs / nhop
    elif extract_type == 'melspec':
      feats, ffs = S.waveform_to_melspec_tf(wav[None], audio_fs, nfft, nhop)[0], audio_fs / nhop
    else:
      raise ValueError()
    f, a = parallel_slice(feats, wav, slice_len, audio_fs, ffs, overlap, pad_end, first_only)
    ex.extend(zip(f, a))
  out = []
  for i in range(0, len(ex) - batch_size + 1, batch_size):
    out.append((np.stack([e[0] for e in ex[i:i + batch_size]]),
                np.stack([e[1] for e in ex[i:i + batch_size]])))
  return out
