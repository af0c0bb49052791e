"""numpy restatement of the reference's STFT / mel feature extractor.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Pinned against the reference's
known-answer constants, tests/test_oracle_spectral.py.

Reference call sites restated here (paths relative to /root/reference):
  advoc/spectral.py:11-41    stft            (numpy / lws path, float64)
  advoc/spectral.py:44-57    lws_hann_default
  advoc/spectral.py:60-83    stft_tf         (TF path, float32)
  advoc/spectral.py:86-94    create_mel_filterbank / create_inverse_mel_filterbank
  advoc/spectral.py:98-154   waveform_to_melspec      (float64)
  advoc/spectral.py:158-227  waveform_to_melspec_tf   (float32)
  models/advoc/spectral_util.py:29-43,52-60  mel <-> mag projections

Third-party algorithms restated (sources are NOT in /root/reference):
  lws 1.2      hann(n, symmetric=True, use_offset=False)[k] = 0.5*(1-cos(2*pi*(k+0.5)/n));
               analysis window sqrt(hann * 2*hop/nfft); stft = frame, window, rfft
               (perfectrec=False: no boundary padding, tail frame zero padded,
               nframes = ceil((n-nfft)/hop)+1).
  TF 1.13      tf.contrib.signal.stft(pad_end=True): nframes = ceil(n/hop), zero tail pad,
               float32 window multiply, rfft(1024) -> complex64.
  librosa 0.6.3 filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm=1): Slaney mel
               scale, triangular weights max(0, min(lower, upper)), area normalised.
"""
import numpy as np


# ----------------------------------------------------------------------------
# window  (advoc/spectral.py:44-57 + lws.hann)
# ----------------------------------------------------------------------------
def lws_hann(n):
  k = np.arange(n, dtype=np.float64)
  return 0.5 * (1.0 - np.cos(2.0 * np.pi * (k + 0.5) / n))


def lws_hann_default(nfft, nhop, dtype=np.float32):
  awin = np.sqrt(lws_hann(nfft) * 2 * nhop / nfft)
  return awin.astype(dtype)


# ----------------------------------------------------------------------------
# framing helpers
# ----------------------------------------------------------------------------
def num_frames_tf(nsamps, nfft, nhop, pad_end=True):
  """tf.contrib.signal.frame frame count (TF 1.13 shape_ops.frame)."""
  if pad_end:
    return -(-nsamps // nhop)
  return max(0, 1 + (nsamps - nfft) // nhop)


def num_frames_lws(nsamps, nfft, nhop):
  """lws.stft frame count with perfectrec=False (tail zero padded)."""
  if nsamps <= 0:
    return 0
  return max(1, int(np.ceil((nsamps - nfft) / float(nhop))) + 1)


def _frame(x, nframes, nfft, nhop):
  need = (nframes - 1) * nhop + nfft if nframes > 0 else 0
  if need > x.shape[0]:
    x = np.concatenate([x, np.zeros(need - x.shape[0], dtype=x.dtype)])
  idx = np.arange(nframes)[:, None] * nhop + np.arange(nfft)[None, :]
  return x[idx]


# ----------------------------------------------------------------------------
# STFT, numpy/lws twin  (advoc/spectral.py:11-41)
# ----------------------------------------------------------------------------
def stft(x, nfft, nhop, pad_end=True):
  """x: [n,1,1] float32 -> [T, nfft//2+1, 1] complex128."""
  nsamps, nfeats, nch = x.shape
  if nfeats != 1:
    raise ValueError()
  if nch != 1:
    raise NotImplementedError('Can only take STFT of monaural signals')
  x = x[:, 0, 0]
  xlen = x.shape[0]
  if pad_end:
    num_frames = int(np.ceil(float(xlen) / nhop) + 1e-6)
    if num_frames > 0:
      pad_amt = (num_frames - 1) * nhop + nfft - xlen
      if pad_amt > 0:
        x = np.pad(x, [[0, pad_amt]], 'constant')
  x = x.astype(np.float64)
  T = num_frames_lws(x.shape[0], nfft, nhop)
  frames = _frame(x, T, nfft, nhop) * lws_hann_default(nfft, nhop, np.float64)[None, :]
  X = np.fft.rfft(frames, n=nfft, axis=1)
  return X.astype(np.complex128)[:, :, np.newaxis]


# ----------------------------------------------------------------------------
# STFT, TF twin (advoc/spectral.py:60-83), float32 arithmetic
# ----------------------------------------------------------------------------
def stft_tf(x, nfft, nhop, pad_end=True):
  """x: [b,n,1,ch] float32 -> [b,T,nfft//2+1,ch] complex64."""
  b, nsamps, nfeats, nch = x.shape
  if nfeats != 1:
    raise ValueError()
  x = np.asarray(x, dtype=np.float32)
  T = num_frames_tf(nsamps, nfft, nhop, pad_end)
  win = lws_hann_default(nfft, nhop, np.float32)
  out = np.zeros((b, T, nfft // 2 + 1, nch), dtype=np.complex64)
  for i in range(b):
    for c in range(nch):
      frames = _frame(x[i, :, 0, c], T, nfft, nhop) * win[None, :]     # float32 multiply
      # pocketfft keeps float32 input in single precision (numpy >= 2) only via scipy;
      # use scipy.fft which preserves float32 -> complex64.
      from scipy import fft as spfft
      out[i, :, :, c] = spfft.rfft(frames.astype(np.float32), n=nfft, axis=1)
  return out


def stft_mag_f64(x, nfft, nhop, pad_end=True):
  """|stft_tf| computed in float64 from float32 samples and the float32 window:
  the 'exact' value of what the TF float32 graph approximates.  Used as the
  high-precision target for the 1e-4 relative-L2 bar."""
  b, nsamps, nfeats, nch = x.shape
  T = num_frames_tf(nsamps, nfft, nhop, pad_end)
  win = lws_hann_default(nfft, nhop, np.float32).astype(np.float64)
  out = np.zeros((b, T, nfft // 2 + 1, nch), dtype=np.float64)
  for i in range(b):
    for c in range(nch):
      frames = _frame(x[i, :, 0, c].astype(np.float64), T, nfft, nhop) * win[None, :]
      out[i, :, :, c] = np.abs(np.fft.rfft(frames, n=nfft, axis=1))
  return out


# ----------------------------------------------------------------------------
# mel filterbank (librosa 0.6.3 filters.mel restated; advoc/spectral.py:86-94)
# ----------------------------------------------------------------------------
_F_SP = 200.0 / 3
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP
_LOGSTEP = np.log(6.4) / 27.0


def hz_to_mel(f):
  f = np.asanyarray(f, dtype=np.float64)
  mels = f / _F_SP
  if f.ndim:
    m = f >= _MIN_LOG_HZ
    mels[m] = _MIN_LOG_MEL + np.log(f[m] / _MIN_LOG_HZ) / _LOGSTEP
  elif f >= _MIN_LOG_HZ:
    mels = _MIN_LOG_MEL + np.log(f / _MIN_LOG_HZ) / _LOGSTEP
  return mels


def mel_to_hz(mels):
  mels = np.asanyarray(mels, dtype=np.float64)
  freqs = _F_SP * mels
  if mels.ndim:
    m = mels >= _MIN_LOG_MEL
    freqs[m] = _MIN_LOG_HZ * np.exp(_LOGSTEP * (mels[m] - _MIN_LOG_MEL))
  elif mels >= _MIN_LOG_MEL:
    freqs = _MIN_LOG_HZ * np.exp(_LOGSTEP * (mels - _MIN_LOG_MEL))
  return freqs


def create_mel_filterbank(sr, n_fft, fmin=0.0, fmax=None, n_mels=128):
  if fmax is None:
    fmax = float(sr) / 2
  n_mels = int(n_mels)
  nbins = int(1 + n_fft // 2)
  weights = np.zeros((n_mels, nbins), dtype=np.float64)
  fftfreqs = np.linspace(0, float(sr) / 2, nbins, endpoint=True)
  mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
  fdiff = np.diff(mel_f)
  ramps = np.subtract.outer(mel_f, fftfreqs)
  for i in range(n_mels):
    lower = -ramps[i] / fdiff[i]
    upper = ramps[i + 2] / fdiff[i + 1]
    weights[i] = np.maximum(0, np.minimum(lower, upper))
  enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
  weights *= enorm[:, np.newaxis]
  return weights


def create_inverse_mel_filterbank(sr, n_fft, fmin=0.0, fmax=None, n_mels=128):
  return np.linalg.pinv(create_mel_filterbank(sr, n_fft, fmin=fmin, fmax=fmax, n_mels=n_mels))


def mel_bin_map(W):
  """(first, last) non-zero FFT bin per mel band -- the 'mel index/bin mapping'
  that north_star requires bit-exact."""
  nz = W > 0
  first = np.argmax(nz, axis=1)
  last = W.shape[1] - 1 - np.argmax(nz[:, ::-1], axis=1)
  return np.stack([first, last], axis=1).astype(np.int32)


# ----------------------------------------------------------------------------
# waveform -> mel (advoc/spectral.py:98-154 numpy f64, :158-227 TF f32)
# ----------------------------------------------------------------------------
def waveform_to_melspec(x, fs, nfft, nhop, mel_min=125, mel_max=7600, mel_num_bins=80,
                        norm_allow_clipping=True, norm_min_level_db=-100, norm_ref_level_db=20):
  if x.dtype != np.float32:
    raise ValueError()
  nsamps, nfeats, nch = x.shape
  if nfeats != 1:
    raise ValueError()
  if nch != 1:
    raise NotImplementedError('Can only extract features from monaural signals')
  X_mag = np.abs(stft(x, nfft, nhop)[:, :, 0])
  W = create_mel_filterbank(fs, nfft, fmin=mel_min, fmax=mel_max, n_mels=mel_num_bins)
  X_mel = np.swapaxes(np.dot(W, X_mag.T), 0, 1)
  min_level = np.exp(norm_min_level_db / 20 * np.log(10))
  X_mel_db = 20 * np.log10(np.maximum(min_level, X_mel)) - norm_ref_level_db
  if not norm_allow_clipping:
    assert X_mel_db.max() <= 0 and X_mel_db.min() - norm_min_level_db >= 0
  X = np.clip((X_mel_db - norm_min_level_db) / -norm_min_level_db, 0, 1)
  return X[:, :, np.newaxis]


def waveform_to_melspec_tf(x, fs, nfft, nhop, mel_min=125, mel_max=7600, mel_num_bins=80,
                           norm_allow_clipping=True, norm_min_level_db=-100,
                           norm_ref_level_db=20):
  """float32 twin: x [b,n,1,ch] f32 -> [b,T,mel,ch] f32."""
  b, nsamps, one, nch = x.shape
  if one != 1:
    raise ValueError()
  if x.dtype != np.float32:
    raise ValueError()
  if not norm_allow_clipping:
    raise NotImplementedError()
  X_mag = np.abs(stft_tf(x, nfft, nhop)).astype(np.float32)            # [b,T,F,ch]
  W = create_mel_filterbank(fs, nfft, fmin=mel_min, fmax=mel_max,
                            n_mels=mel_num_bins).astype(np.float32)
  X_mel = np.einsum('btfc,mf->btmc', X_mag, W).astype(np.float32)
  min_level = np.float32(np.exp(norm_min_level_db / 20 * np.log(10)))
  log10 = np.log(np.maximum(min_level, X_mel)) / np.log(np.float32(10))
  X_mel_db = np.float32(20) * log10.astype(np.float32) - np.float32(norm_ref_level_db)
  X = np.clip((X_mel_db - norm_min_level_db) / -norm_min_level_db, 0, 1)
  return X.astype(np.float32)


def waveform_to_r9y9_melspec(x, fs=22050):
  return waveform_to_melspec(x, fs=fs, nfft=1024, nhop=256)


def waveform_to_r9y9_melspec_tf(x, fs=22050):
  return waveform_to_melspec_tf(x, fs=fs, nfft=1024, nhop=256)


def waveform_to_tacotron2_melspec(x):
  return waveform_to_melspec(x, fs=24000, nfft=1200, nhop=300, norm_min_level_db=-40)


# ----------------------------------------------------------------------------
# SpectralUtil projections (models/advoc/spectral_util.py:29-43, 52-60)
# ----------------------------------------------------------------------------
def mag_to_mel_linear_spec(mag, W32):
  """mag [B,T,513,1] f32, W32 [80,513] f32 -> [B,T,80,1] f32 (linear amplitude)."""
  return np.tensordot(mag[:, :, :, 0], W32.T, axes=1)[..., np.newaxis].astype(np.float32)


def mel_linear_to_mag_spec(mel, Winv32):
  """mel [B,T,80,1] f32, Winv32 [513,80] f32 -> [B,T,513,1] f32.  No >=0 clamp."""
  return np.tensordot(mel[:, :, :, 0], Winv32.T, axes=1)[..., np.newaxis].astype(np.float32)


def tacotron_mel_to_mag(X_mel_dbnorm, invmeltrans):
  """scripts/spectrogram_advoc.py:15-22 / spectral_util.py:52-60 (float64, no clip)."""
  norm_min_level_db = -100
  norm_ref_level_db = 20
  X_mel_db = (X_mel_dbnorm * -norm_min_level_db) + norm_min_level_db
  X_mel = np.power(10, (X_mel_db + norm_ref_level_db) / 20)
  return np.dot(X_mel, invmeltrans.T)


# ----------------------------------------------------------------------------
# inversion: iSTFT + Griffin-Lim  (advoc/spectral.py:294-311; lws 1.2 istft restated)
# ----------------------------------------------------------------------------
def synth_window(awin, nhop):
  """lws.synthwindow: swin = awin / sum_q (awin * awin)[k + q*hop] -- the synthesis window that
  makes overlap-add of analysis-windowed frames the identity.  For the lws default
  sqrt(hann * 2*hop/nfft) with hop = nfft/4 the denominator is exactly 1 (swin == awin)."""
  awin = np.asarray(awin, dtype=np.float64)
  n = awin.shape[0]
  q = -(-n // nhop)
  sq = np.zeros(q * nhop, dtype=np.float64)
  sq[:n] = awin * awin
  den = np.tile(sq.reshape(q, nhop).sum(axis=0), q)[:n]
  return awin / den


def istft(X, nfft, nhop):
  """lws(nfft, nhop, perfectrec=False).istft: X [T, nfft//2+1] complex -> [(T-1)*hop + nfft]
  float64 (irfft per frame, synthesis window, overlap-add; no boundary trimming)."""
  X = np.asarray(X)
  T = X.shape[0]
  if T == 0:
    return np.zeros(0, dtype=np.float64)
  swin = synth_window(lws_hann_default(nfft, nhop, np.float64), nhop)
  frames = np.fft.irfft(X.astype(np.complex128), n=nfft, axis=1) * swin[None, :]
  out = np.zeros((T - 1) * nhop + nfft, dtype=np.float64)
  for t in range(T):
    out[t * nhop:t * nhop + nfft] += frames[t]
  return out


def _stft_lws_1d(x, nfft, nhop):
  T = num_frames_lws(x.shape[0], nfft, nhop)
  frames = _frame(x.astype(np.float64), T, nfft, nhop) * lws_hann_default(nfft, nhop, np.float64)[None, :]
  return np.fft.rfft(frames, n=nfft, axis=1)


def magspec_to_waveform_griffin_lim(X_mag, nfft, nhop, ngl=60, angles0=None):
  """advoc/spectral.py:294-311.  X_mag [T, bins, 1] -> float32 [(T-1)*hop + nfft, 1, 1].
  The reference draws the initial phases from numpy's GLOBAL generator
  (np.random.rand(*X_mag.shape)); `angles0` (uniform [0,1) array) overrides that for tests."""
  nsamps, nbins, nch = X_mag.shape
  if nch != 1:
    raise NotImplementedError('Can only invert monaural signals')
  X_mag = X_mag[:, :, 0]
  u = np.random.rand(*X_mag.shape) if angles0 is None else np.asarray(angles0, dtype=np.float64)
  angles = np.exp(2j * np.pi * u)
  X_complex = np.abs(X_mag).astype(np.complex128)
  x_gl = istft(X_complex * angles, nfft, nhop)
  for _ in range(ngl):
    angles = np.exp(1j * np.angle(_stft_lws_1d(x_gl, nfft, nhop)))
    x_gl = istft(X_complex * angles, nfft, nhop)
  return x_gl[:, np.newaxis, np.newaxis].astype(np.float32)
