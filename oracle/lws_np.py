"""Local Weighted Sums (LWS) phase reconstruction -- CPU oracle (TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this).

PARITY UNPINNED.  The reference calls the third-party C++ library lws 1.2 (`lws.lws(nfft, nhop, mode='speech',
perfectrec=False).run_lws(X_mag)` at /root/reference/advoc/spectral.py:314-326 and
models/advoc/spectral_util.py:45-50); the library is not part of /root/reference and cannot be installed here, so this
file restates the PUBLISHED algorithm (Le Roux, Kameoka, Ono, Sagayama, "Fast signal reconstruction from magnitude STFT
spectrogram based on spectrogram consistency", DAFx 2010, and "Phase initialization schemes for faster spectrogram-
consistency-based signal reconstruction", ASJ 2010), in the three stages run_lws chains:

  consistency    a complex spectrogram X is the STFT of some signal iff X = P X, P = STFT o iSTFT.  P is a small 2-D
                 convolution: (P X)[t, f] = sum_{q, p} K_q(f, p) X[t + q, f + p] with |q| < Q = nfft / nhop frames and a
                 kernel that decays fast in |p|; LWS truncates it to |p| < L (L = 5) and drops the centre term:
                     X[t, f] <- |A[t, f]| * phase( sum_{(q, p) != (0, 0)} K_q(f, p) X[t + q, f + p] )
                 with K_q(f, p) = exp(-2 pi i (f + p) q nhop / nfft) / nfft * sum_n awin[n] swin[n - q nhop] exp(+2 pi i p n / nfft)
                 for the frame-local phase convention of lws's STFT (oracle/spectral_np.py:stft).
  1 no-future    frames are visited in time order; a new frame starts EMPTY and its bins are set from the frames before
                 it (q <= 0) and the bins of its own already set -- first the bins above the mean magnitude, then all
                 (zero phase where nothing is known yet);
  2 online       ... then refined `online_iterations` times with a look-ahead of `look_ahead` frames, which at that
                 point hold their no-future initialisation;
  3 batch        `batch_iterations` sweeps over the whole spectrogram; sweep i only touches bins whose magnitude
                 exceeds alpha * exp(-beta * i^gamma) times the mean magnitude (strong bins settle first).
The frame-level updates are Jacobi updates (all bins of a frame / of the spectrogram from the previous iterate), which is
what the GPU kernels do (advoc_amd/csrc/lws.hip); lws itself updates in place bin by bin.
"""
import numpy as np

from oracle import spectral_np as S


def projection_kernel(awin, swin, nhop, L=5):
  """K[q + Q - 1, p + L - 1] WITHOUT the exp(-2 pi i (f + p) q nhop / nfft) factor: alpha_q(p) =
  1 / nfft * sum_n awin[n] swin[n - q nhop] exp(2 pi i p n / nfft), complex128 [2Q - 1, 2L - 1]."""
  awin = np.asarray(awin, dtype=np.float64)
  swin = np.asarray(swin, dtype=np.float64)
  nfft = awin.shape[0]
  Q = -(-nfft // nhop)
  n = np.arange(nfft)
  K = np.zeros((2 * Q - 1, 2 * L - 1), dtype=np.complex128)
  for q in range(-(Q - 1), Q):
    sh = np.zeros(nfft)
    lo, hi = max(0, q * nhop), min(nfft, nfft + q * nhop)
    if hi > lo:
      sh[lo:hi] = swin[lo - q * nhop:hi - q * nhop]
    prod = awin * sh
    for p in range(-(L - 1), L):
      K[q + Q - 1, p + L - 1] = np.sum(prod * np.exp(2j * np.pi * p * n / nfft)) / nfft
  return K


def _extended(X, L):
  """[T, F] one-sided spectrum -> [T, F + 2 (L - 1)] with the conjugate-symmetric bins below 0 and above nfft / 2."""
  lo = np.conj(X[:, L - 1:0:-1])
  hi = np.conj(X[:, -2:-(L + 1):-1])
  return np.concatenate([lo, X, hi], axis=1)


def local_sums(X, K, nhop, nfft, qs, t_lo=0, t_hi=None, include_centre=False):
  """sum_{q in qs, p} K_q(f, p) X[t + q, f + p] for frames t_lo <= t < t_hi (zeros outside the spectrogram)."""
  T, F = X.shape
  Q = (K.shape[0] + 1) // 2
  L = (K.shape[1] + 1) // 2
  t_hi = T if t_hi is None else t_hi
  Xe = _extended(X, L)
  out = np.zeros((t_hi - t_lo, F), dtype=np.complex128)
  f = np.arange(F)
  for q in qs:
    src_lo, src_hi = t_lo + q, t_hi + q
    a, b = max(src_lo, 0), min(src_hi, T)
    if b <= a:
      continue
    for p in range(-(L - 1), L):
      if q == 0 and p == 0 and not include_centre:
        continue
      rot = np.exp(-2j * np.pi * ((f + p) % nfft) * q * nhop / nfft)
      out[a - src_lo:b - src_lo] += K[q + Q - 1, p + L - 1] * rot[None, :] * Xe[a:b, f + p + L - 1]
  return out


def _with_phase_of(mag, Z, old):
  """mag * Z / |Z|; keeps `old` where Z == 0."""
  a = np.abs(Z)
  return np.where(a > 0, mag * Z / np.where(a > 0, a, 1.0), old)


NOFUTURE_THRESHOLDS = (1.0, 0.0)


def run_lws(X, nfft, nhop, L=5, look_ahead=3, nofuture_thresholds=NOFUTURE_THRESHOLDS, online_iterations=10,
            online_alpha=1.0, online_beta=0.1, batch_iterations=100, batch_alpha=100.0, batch_beta=0.1, batch_gamma=1.0):
  """[T, nfft // 2 + 1] magnitudes (real input: phases start from nothing) or a complex spectrogram (its phases are the
  starting point, as lws.run_lws treats complex input -- the reference's own test hands it the true STFT,
  tests/test_spectral.py:184,190) -> complex128 spectrogram with the estimated phases.  Thresholds are multiples of the
  mean magnitude."""
  X = np.asarray(X)
  use_init = np.iscomplexobj(X)
  A = np.abs(X).astype(np.float64)
  T, F = A.shape
  awin = S.lws_hann_default(nfft, nhop, np.float64)
  swin = S.synth_window(awin, nhop)
  K = projection_kernel(awin, swin, nhop, L)
  Q = (K.shape[0] + 1) // 2
  past = list(range(-(Q - 1), 1))
  full = list(range(-(Q - 1), Q))
  ref = A.mean()
  Xc = X.astype(np.complex128) if use_init else np.zeros((T, F), dtype=np.complex128)

  def init_frame(t):
    if use_init:
      return
    for thr in nofuture_thresholds:
      Z = local_sums(Xc, K, nhop, nfft, past, t, t + 1)[0]
      m = (A[t] > thr * ref) if thr > 0 else np.ones(F, dtype=bool)
      Xc[t] = np.where(m, _with_phase_of(A[t], Z, A[t].astype(np.complex128)), Xc[t])

  if use_init:
    keep = Xc.copy()       # frames ahead of the sweep count as "not initialised yet", as in the GPU ring
  last = -1
  for t0 in range(min(look_ahead, T - 1) + 1):
    init_frame(t0)
    last = t0
  for t in range(T):
    ta = t + look_ahead
    if ta < T and ta > look_ahead:
      init_frame(ta)
      last = ta
    for i in range(online_iterations):
      thr = online_alpha * np.exp(-online_beta * i) * ref if online_iterations > 1 else 0.0
      view = Xc
      if use_init and last < T - 1:
        view = Xc.copy()
        view[last + 1:] = 0
      Z = local_sums(view, K, nhop, nfft, full, t, t + 1)[0]
      Xc[t] = np.where(A[t] > thr, _with_phase_of(A[t], Z, Xc[t]), Xc[t])
  for i in range(batch_iterations):
    thr = batch_alpha * np.exp(-batch_beta * float(i) ** batch_gamma) * ref
    Z = local_sums(Xc, K, nhop, nfft, full)
    Xc = np.where(A > thr, _with_phase_of(A, Z, Xc), Xc)
  return Xc


def magspec_to_waveform_lws(X_mag, nfft, nhop, **kw):
  """advoc/spectral.py:314-326: [T, bins, 1] -> float32 [(T - 1) nhop + nfft, 1, 1]."""
  nsamps, nbins, nch = X_mag.shape
  if nch != 1:
    raise NotImplementedError('Can only invert monaural signals')
  X = run_lws(np.asarray(X_mag)[:, :, 0], nfft, nhop, **kw)
  return S.istft(X, nfft, nhop)[:, np.newaxis, np.newaxis].astype(np.float32)
