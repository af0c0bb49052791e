"""STFT / mel feature extractor on MI355X -- drop-in for the reference's ``advoc.spectral``.

Same function names, argument names, defaults, shape conventions and error
behaviour as /root/reference/advoc/spectral.py; the arithmetic runs in the HIP
kernels of libadvoc_hip.so (csrc/stft.hip, csrc/features.hip).

Two calling conventions, as in the reference:
  * "numpy" API (``stft``, ``waveform_to_melspec`` ...): ``[n, 1, 1] float32``
    numpy in, ``[T, F, 1]`` complex128 / float64 numpy out.  The reference runs
    these through lws in float64 on the host (spectral.py:11-41); here they are
    computed by the same float32 HIP kernels and widened on return
    (|rel err| ~1e-7 versus the float64 path, far inside the 1e-4 parity bar).
  * tensor API (``stft_tf``, ``waveform_to_melspec_tf`` ...): ``[b, n, 1, ch]``
    float32 in, ``[b, T, F, ch]`` out.  The reference returns lazy TF tensors;
    these return eager ``torch.Tensor``s resident in HBM (numpy input is
    uploaded first).

The mel filterbank and its pseudo-inverse are float64 host constants computed
once (reference: librosa.filters.mel + np.linalg.pinv, spectral.py:86-94).

Phase reconstruction (spectral.py:294-326): Griffin-Lim, LWS and the inverse STFT run on the GPU
(advoc_istft_f32, advoc_lws_* & friends); LWS restates the published algorithm of the third-party lws library
(parity unpinned).
"""
import ctypes
import math
from functools import lru_cache

import numpy as np
import torch

from advoc_amd import _lib

# ---------------------------------------------------------------------------------------------
# constants built on the host once
# ---------------------------------------------------------------------------------------------


@lru_cache(maxsize=8)
def _lws_window_f64(nfft, nhop):
  # lws.hann(nfft, symmetric=True, use_offset=False): half-sample-offset Hann;
  # analysis window sqrt(hann * 2 * hop / nfft)   (reference spectral.py:55-56)
  phase = (np.arange(nfft, dtype=np.float64) + 0.5) * (2.0 * np.pi / nfft)
  return np.sqrt((0.5 - 0.5 * np.cos(phase)) * 2 * nhop / nfft)


def lws_hann_default(nfft, nhop, dtype=torch.float32):
  """Default LWS sqrt-Hann analysis window, shape [nfft] (reference spectral.py:44-57)."""
  return torch.from_numpy(_lws_window_f64(int(nfft), int(nhop)).copy()).to(dtype)


_dev_cache = {}


def _device_const(key, make):
  dev = _lib.device()
  k = (key, dev.index)
  t = _dev_cache.get(k)
  if t is None:
    t = make().to(dev).contiguous()
    _dev_cache[k] = t
  return t


def _device_window(nfft, nhop):
  return _device_const(('win', nfft, nhop), lambda: lws_hann_default(nfft, nhop, torch.float32))


def _device_twiddle(nfft):
  def make():
    import ctypes
    host = torch.empty(2 * nfft, dtype=torch.float32)
    _lib.check(_lib.load().advoc_stft_twiddle_host(ctypes.c_void_p(host.data_ptr()), nfft),
               'advoc_stft_twiddle_host')
    return host
  return _device_const(('twiddle', nfft), make)


def _slaney_hz_to_mel(hz):
  hz = np.atleast_1d(np.asarray(hz, dtype=np.float64))
  lin = hz * (3.0 / 200.0)
  log = 15.0 + np.log(np.maximum(hz, 1e-300) / 1000.0) * (27.0 / np.log(6.4))
  return np.where(hz >= 1000.0, log, lin)


def _slaney_mel_to_hz(mel):
  mel = np.atleast_1d(np.asarray(mel, dtype=np.float64))
  lin = mel * (200.0 / 3.0)
  log = 1000.0 * np.exp((np.log(6.4) / 27.0) * (mel - 15.0))
  return np.where(mel >= 15.0, log, lin)


@lru_cache(maxsize=4)
def create_mel_filterbank(*args, **kwargs):
  """Slaney-scale, area-normalised triangular mel filterbank, float64 [n_mels, 1+n_fft/2].

  Call as the reference does: ``create_mel_filterbank(fs, nfft, fmin=, fmax=, n_mels=)``
  (spectral.py:86-88 -> librosa 0.6.3 ``filters.mel(sr, n_fft, n_mels=128, fmin=0.0,
  fmax=None, htk=False, norm=1)``).
  """
  def _sig(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    return sr, n_fft, n_mels, fmin, fmax
  sr, n_fft, n_mels, fmin, fmax = _sig(*args, **kwargs)
  if fmax is None:
    fmax = float(sr) / 2
  n_mels = int(n_mels)
  nbins = 1 + int(n_fft) // 2
  bin_hz = np.linspace(0, float(sr) / 2, nbins, endpoint=True)
  lo, hi = _slaney_hz_to_mel(fmin)[0], _slaney_hz_to_mel(fmax)[0]
  edges = _slaney_mel_to_hz(np.linspace(lo, hi, n_mels + 2))           # band edges in Hz
  width = np.diff(edges)
  dist = edges[:, None] - bin_hz[None, :]                              # [n_mels+2, nbins]
  rising = -dist[:n_mels] / width[:n_mels, None]
  falling = dist[2:] / width[1:, None]
  tri = np.maximum(0, np.minimum(rising, falling))
  tri *= (2.0 / (edges[2:] - edges[:n_mels]))[:, None]
  return tri


@lru_cache(maxsize=4)
def create_inverse_mel_filterbank(*args, **kwargs):
  """np.linalg.pinv of the mel filterbank, float64 [1+n_fft/2, n_mels] (spectral.py:91-94)."""
  return np.linalg.pinv(create_mel_filterbank(*args, **kwargs))


def mel_bin_map(W):
  """(first, last) non-zero FFT bin of every mel band, int32 [n_mels, 2]."""
  nz = np.asarray(W) > 0
  first = nz.argmax(axis=1)
  last = nz.shape[1] - 1 - nz[:, ::-1].argmax(axis=1)
  return np.stack([first, last], axis=1).astype(np.int32)


# ---------------------------------------------------------------------------------------------
# device kernels
# ---------------------------------------------------------------------------------------------
def _frames_pad_end(nsamps, nhop):
  return -(-nsamps // nhop)


def _frames_tf_nopad(nsamps, nfft, nhop):
  return max(0, 1 + (nsamps - nfft) // nhop)


def _frames_lws(nsamps, nfft, nhop):
  if nsamps <= 0:
    return 0
  return max(1, -(-(nsamps - nfft) // nhop) + 1)


def _to_device_f32(x):
  if isinstance(x, np.ndarray):
    x = torch.from_numpy(np.ascontiguousarray(x))
  if not isinstance(x, torch.Tensor):
    raise ValueError('expected a numpy array or torch tensor')
  return x.to(_lib.device())


def _run_stft(wav2d, nfft, nhop, nframes, complex_out):
  """wav2d: [clips, n] float32 contiguous on device -> [clips, T, F] (mag) or [clips,T,F,2]."""
  clips, n = wav2d.shape
  nbins = nfft // 2 + 1
  shape = (clips, nframes, nbins, 2) if complex_out else (clips, nframes, nbins)
  out = torch.empty(shape, dtype=torch.float32, device=wav2d.device)
  fn = _lib.load().advoc_stft_c64 if complex_out else _lib.load().advoc_stft_mag_f32
  if nframes == 0 or clips == 0:
    return out
  if nfft != 1024 or nhop % 2 or nhop > 4096:      # outside the fused kernel: windowed DFT as a matmul
    return _run_stft_generic(wav2d, nfft, nhop, nframes, complex_out, out)
  win = _device_window(nfft, nhop)
  _lib.check(fn(_lib.ptr(wav2d), clips, n, _lib.ptr(win), _lib.ptr(_device_twiddle(nfft)), nfft, nhop,
                nframes, _lib.ptr(out), _lib.stream()), 'advoc_stft')
  return out


def _dft_basis(nfft, nhop):
  """[2*bins, nfft] float32: rows (2k, 2k+1) = window * (cos, -sin)(2 pi k n / nfft), float64 on the host."""
  def make():
    n = np.arange(nfft, dtype=np.float64)
    k = np.arange(nfft // 2 + 1, dtype=np.float64)
    ang = 2.0 * np.pi * np.outer(k, n) / nfft
    w = _lws_window_f64(nfft, nhop).astype(np.float32).astype(np.float64)     # the fp32 window, as TF applies it
    basis = np.empty((2 * k.shape[0], nfft), dtype=np.float64)
    basis[0::2] = np.cos(ang) * w[None, :]
    basis[1::2] = -np.sin(ang) * w[None, :]
    return torch.from_numpy(basis.astype(np.float32))
  return _device_const(('dft', nfft, nhop), make)


def _run_stft_generic(wav2d, nfft, nhop, nframes, complex_out, out):
  """Any frame length (the Tacotron-2 preset uses nfft 1200 / hop 300, spectral.py:241-247): the
  windowed DFT is one [frames, nfft] x [nfft, 2*bins] product on advoc_matmul_nt_f32.  Not a hot
  path -- training and vocoding run the fused 1024-point kernel."""
  clips, n = wav2d.shape
  need = (nframes - 1) * nhop + nfft
  if need > n:
    wav2d = torch.nn.functional.pad(wav2d, (0, need - n))
  frames = wav2d.unfold(1, nfft, nhop)[:, :nframes].contiguous()          # data movement only
  spec = matmul_last(frames, _dft_basis(nfft, nhop))                       # [clips, T, 2*bins]
  if complex_out:
    out.copy_(spec.reshape(out.shape))
    return out
  _lib.check(_lib.load().advoc_cabs_f32(_lib.ptr(spec), _lib.ptr(out), out.numel(), _lib.stream()),
             'advoc_cabs_f32')
  return out


def _clips_first(x):
  """[b, n, 1, ch] -> contiguous [b*ch, n]."""
  b, n, _, ch = x.shape
  if ch == 1:
    return x.reshape(b, n).contiguous()
  return x[:, :, 0, :].permute(0, 2, 1).reshape(b * ch, n).contiguous()


def _channels_last(y, b, ch):
  """[b*ch, T, F(, 2)] -> [b, T, F, ch(, 2)]."""
  if ch == 1:
    return y.unsqueeze(3)
  y = y.reshape((b, ch) + tuple(y.shape[1:]))
  return y.movedim(1, 3).contiguous()


def stft_magnitude(x, nfft, nhop, pad_end=True):
  """|STFT| straight from the fused HIP kernel: [b,n,1,ch] f32 -> [b,T,F,ch] f32.

  This is what the loader's magspec branch (reference loader.py:116-128:
  ``tf.abs(stft_tf(...))``) runs; the complex spectrum is never materialised.
  """
  x = _to_device_f32(x)
  b, n, nfeats, ch = x.shape
  if nfeats != 1:
    raise ValueError()
  if x.dtype != torch.float32:
    raise ValueError()
  T = _frames_pad_end(n, nhop) if pad_end else _frames_tf_nopad(n, nfft, nhop)
  mag = _run_stft(_clips_first(x), nfft, nhop, T, complex_out=False)
  return _channels_last(mag, b, ch)


def stft_tf(x, nfft, nhop, pad_end=True):
  """Short-time Fourier transform of a waveform batch (reference spectral.py:60-83).

  Args:
    x: float32 [b, nsamps, 1, nch] (torch tensor, or numpy which is uploaded).
  Returns:
    complex64 torch tensor [b, ntsteps, nfft // 2 + 1, nch] in HBM.
  """
  x = _to_device_f32(x)
  b, n, nfeats, ch = x.shape
  if nfeats != 1:
    raise ValueError()
  if x.dtype != torch.float32:
    raise ValueError()
  T = _frames_pad_end(n, nhop) if pad_end else _frames_tf_nopad(n, nfft, nhop)
  X = _run_stft(_clips_first(x), nfft, nhop, T, complex_out=True)
  X = torch.view_as_complex(X)
  return _channels_last(X, b, ch)


def stft(x, nfft, nhop, pad_end=True):
  """STFT of one mono waveform, numpy API (reference spectral.py:11-41).

  Args:
    x: nd-array float32 [nsamps, 1, 1].
  Returns:
    nd-array complex128 [ntsteps, nfft // 2 + 1, 1].
  """
  nsamps, nfeats, nch = x.shape
  if nfeats != 1:
    raise ValueError()
  if nch != 1:
    raise NotImplementedError('Can only take STFT of monaural signals')
  if pad_end:
    T = int(np.ceil(float(nsamps) / nhop) + 1e-6)
  else:
    T = _frames_lws(nsamps, nfft, nhop)   # lws zero-pads the last partial frame
  wav = _to_device_f32(np.asarray(x[:, 0, 0], dtype=np.float32)).reshape(1, nsamps)
  X = torch.view_as_complex(_run_stft(wav, nfft, nhop, T, complex_out=True))[0]
  return X.cpu().numpy().astype(np.complex128)[:, :, np.newaxis]


def matmul_last(x, w_dev):
  """x [..., K] (device f32) times w_dev[N, K]^T -> [..., N] through advoc_matmul_nt_f32."""
  K = x.shape[-1]
  N = w_dev.shape[0]
  x2 = x.reshape(-1, K).contiguous()
  out = torch.empty((x2.shape[0], N), dtype=torch.float32, device=x.device)
  _lib.check(_lib.load().advoc_matmul_nt_f32(_lib.ptr(x2), _lib.ptr(w_dev), _lib.ptr(out),
                                             x2.shape[0], K, N, _lib.stream()),
             'advoc_matmul_nt_f32')
  return out.reshape(tuple(x.shape[:-1]) + (N,))


def mel_and_inverse(mag, meltrans_dev, invmeltrans_dev, packed=None):
  """mag [..., bins] (device f32) -> (mag W^T [..., n_mels], (mag W^T) P^T [..., bins]) in one launch
  (advoc_mel_pinv_f32): the chain models/advoc/train_evaluate.py:55-56 builds from spectral_util.py:29-43.  packed:
  (runs int32 [n_mels, 2], packed weights f32, pseudo-inverse transposed [n_mels, bins]) on the device, as
  pack_filterbank makes them (made here from the dense matrices when not given).  Falls back to two projections for
  shapes the fused kernel does not take."""
  bins, n_mels = mag.shape[-1], meltrans_dev.shape[0]
  if packed is None:
    packed = pack_filterbank(meltrans_dev.cpu().numpy(), invmeltrans_dev.cpu().numpy(), mag.device)
  runs, wp, inv_t = packed
  x2 = mag.reshape(-1, bins).contiguous()
  mel = torch.empty((x2.shape[0], n_mels), dtype=torch.float32, device=mag.device)
  inv = torch.empty((x2.shape[0], bins), dtype=torch.float32, device=mag.device)
  rc = _lib.load().advoc_mel_pinv_f32(_lib.ptr(x2), _lib.ptr(wp), _lib.ptr(runs), _lib.ptr(inv_t), _lib.ptr(mel),
                                      _lib.ptr(inv), x2.shape[0], bins, n_mels, int(wp.numel()), _lib.stream())
  if rc == _lib.ERR_UNSUPPORTED:
    mel = matmul_last(mag, meltrans_dev)
    return mel, matmul_last(mel, invmeltrans_dev)
  _lib.check(rc, 'advoc_mel_pinv_f32')
  lead = tuple(mag.shape[:-1])
  return mel.reshape(lead + (n_mels,)), inv.reshape(lead + (bins,))


def pack_filterbank(meltrans_np, invmeltrans_np, device):
  """(runs, packed weights, transposed pseudo-inverse) of advoc_mel_pinv_f32 as device tensors: every row of the
  filterbank [n_mels, bins] as its run of non-zero weights, padded with zeros to a multiple of 4."""
  w = np.asarray(meltrans_np, dtype=np.float32)
  runs = band_runs(w)
  parts = []
  for m, (lo, hi) in enumerate(runs):
    seg = w[m, lo:hi]
    parts.append(np.concatenate([seg, np.zeros((-len(seg)) % 4, dtype=np.float32)]))
  wp = np.concatenate(parts) if parts else np.zeros(0, dtype=np.float32)
  inv_t = np.ascontiguousarray(np.asarray(invmeltrans_np, dtype=np.float32).T)
  return (torch.from_numpy(runs).to(device), torch.from_numpy(wp).to(device), torch.from_numpy(inv_t).to(device))


def pack_inverse_pairs(invmeltrans_np, device):
  """The pseudo-inverse [bins, n_mels] pre-split for the f16 matrix cores of advoc_stft_mel_pinv_f32: every row n under
  its own power of two 2^s(n) (largest |P[n, :]| in [2^13, 2^14)) as fp16 pairs h0 + h1 (round to nearest), laid out
  [bins / 32][n_mels / 16][plane][lane = 32 half + l32][8] in MFMA operand order, plus unscale[n] = 2^-s(n).  Same
  arithmetic as the per-launch split of mel_pinv_kernel (csrc/melpinv.hip), done once on the host."""
  P = np.asarray(invmeltrans_np, dtype=np.float32)
  bins, n_mels = P.shape
  nb, steps = (bins + 31) // 32, n_mels // 16
  Pp = np.zeros((nb * 32, n_mels), dtype=np.float32)
  Pp[:bins] = P
  amax = np.abs(Pp).max(axis=1)
  _, ex = np.frexp(amax)                                # amax = m 2^ex, m in [0.5, 1): biased exponent - 127 = ex - 1
  sh = np.clip(14 - ex, -120, 120).astype(np.float64)
  up = np.where(amax > 0, np.exp2(sh), 1.0).astype(np.float32)
  a = (Pp * up[:, None]).astype(np.float32)             # exact: a power of two
  h0 = a.astype(np.float16)
  h1 = (a - h0.astype(np.float32)).astype(np.float16)
  out = np.zeros((nb, steps, 2, 64, 8), dtype=np.float16)
  for pl, h in ((0, h0), (1, h1)):
    # h[n = 32 nb + l32, k = 16 st + 8 half + i] -> [nb, st, half, l32, i]
    v = h.reshape(nb, 32, steps, 2, 8).transpose(0, 2, 3, 1, 4)
    out[:, :, pl] = v.reshape(nb, steps, 64, 8)
  return (torch.from_numpy(out.view(np.int16)).to(device).contiguous(), torch.from_numpy((1.0 / up).astype(np.float32)).to(device))


def stft_mel_inverse(wav2d, nfft, nhop, nframes, packed, pairs):
  """wav2d [clips, n] (device f32) -> (|STFT| [clips, T, bins], mel [clips, T, n_mels], inv [clips, T, bins]) in ONE
  launch (advoc_stft_mel_pinv_f32, csrc/extract.hip); None when the shapes are outside the fused kernel."""
  runs, wp, inv_t = packed
  tab, unscale = pairs
  clips, n = wav2d.shape
  n_mels, bins = inv_t.shape
  mag = torch.empty((clips, nframes, bins), dtype=torch.float32, device=wav2d.device)
  mel = torch.empty((clips, nframes, n_mels), dtype=torch.float32, device=wav2d.device)
  inv = torch.empty((clips, nframes, bins), dtype=torch.float32, device=wav2d.device)
  if nfft != 1024:
    return None
  rc = _lib.load().advoc_stft_mel_pinv_f32(
      _lib.ptr(wav2d), clips, n, _lib.ptr(_device_window(nfft, nhop)), _lib.ptr(_device_twiddle(nfft)), nfft, nhop, nframes,
      _lib.ptr(wp), _lib.ptr(runs), int(wp.numel()), bins, n_mels, _lib.ptr(tab), _lib.ptr(unscale), _lib.ptr(mag),
      _lib.ptr(mel), _lib.ptr(inv), _lib.stream())
  if rc == _lib.ERR_UNSUPPORTED:
    return None
  _lib.check(rc, 'advoc_stft_mel_pinv_f32')
  return mag, mel, inv


def band_runs(meltrans_np):
  """[n_mels, 2] int32: first bin and one past the last bin with a non-zero weight in each filterbank row."""
  out = np.zeros((meltrans_np.shape[0], 2), dtype=np.int32)
  for m, row in enumerate(np.asarray(meltrans_np)):
    nz = np.flatnonzero(row)
    if len(nz):
      out[m] = (nz[0], nz[-1] + 1)
  return out


def _device_melbank(fs, nfft, mel_min, mel_max, mel_num_bins):
  key = ('mel', fs, nfft, mel_min, mel_max, mel_num_bins)
  return _device_const(key, lambda: torch.from_numpy(create_mel_filterbank(
      fs, nfft, fmin=mel_min, fmax=mel_max, n_mels=mel_num_bins).astype(np.float32)))


def waveform_to_melspec_tf(
    x,
    fs,
    nfft,
    nhop,
    mel_min=125,
    mel_max=7600,
    mel_num_bins=80,
    norm_allow_clipping=True,
    norm_min_level_db=-100,
    norm_ref_level_db=20):
  """Waveform batch -> dB-normalised mel spectrogram (reference spectral.py:158-227).

  Args:
    x: float32 [b, nsamps, 1, nch].
  Returns:
    float32 torch tensor [b, ntsteps, mel_num_bins, nch] in HBM, values in [0, 1].
  """
  x = _to_device_f32(x)
  b, n, one, ch = x.shape
  if one != 1:
    raise ValueError()
  if x.dtype != torch.float32:
    raise ValueError()
  if not norm_allow_clipping:
    raise NotImplementedError()
  T = _frames_pad_end(n, nhop)
  mag = _run_stft(_clips_first(x), nfft, nhop, T, complex_out=False)       # [b*ch, T, F]
  mel = matmul_last(mag, _device_melbank(fs, nfft, mel_min, mel_max, mel_num_bins))
  min_level = float(np.float32(np.exp(norm_min_level_db / 20 * np.log(10))))
  _lib.check(_lib.load().advoc_mel_dbnorm_f32(_lib.ptr(mel), mel.numel(), min_level,
                                              float(norm_ref_level_db), float(norm_min_level_db),
                                              _lib.stream()), 'advoc_mel_dbnorm_f32')
  return _channels_last(mel, b, ch)


def waveform_to_melspec(
    x,
    fs,
    nfft,
    nhop,
    mel_min=125,
    mel_max=7600,
    mel_num_bins=80,
    norm_allow_clipping=True,
    norm_min_level_db=-100,
    norm_ref_level_db=20):
  """One mono waveform -> mel spectrogram, numpy API (reference spectral.py:98-154).

  Args:
    x: nd-array float32 [nsamps, 1, 1].
  Returns:
    nd-array float64 [ntsteps, mel_num_bins, 1].
  """
  if x.dtype != np.float32:
    raise ValueError()
  nsamps, nfeats, nch = x.shape
  if nfeats != 1:
    raise ValueError()
  if nch != 1:
    raise NotImplementedError('Can only extract features from monaural signals')
  y = waveform_to_melspec_tf(x[np.newaxis], fs, nfft, nhop, mel_min=mel_min, mel_max=mel_max,
                             mel_num_bins=mel_num_bins, norm_allow_clipping=True,
                             norm_min_level_db=norm_min_level_db,
                             norm_ref_level_db=norm_ref_level_db)
  y = y[0].cpu().numpy().astype(np.float64)
  if not norm_allow_clipping:
    # the reference asserts on the dB values BEFORE the clip, bounds inclusive (spectral.py:150-151): recompute
    # them from the linear mel spectrogram (device STFT + projection, dB on the host)
    xd = _to_device_f32(x[np.newaxis])
    T = _frames_pad_end(nsamps, nhop)
    mag = _run_stft(_clips_first(xd), nfft, nhop, T, complex_out=False)
    mel = matmul_last(mag, _device_melbank(fs, nfft, mel_min, mel_max, mel_num_bins)).cpu().numpy().astype(np.float64)
    min_level = np.exp(norm_min_level_db / 20 * np.log(10))
    db = 20 * np.log10(np.maximum(min_level, mel)) - norm_ref_level_db
    assert db.max() <= 0 and db.min() - norm_min_level_db >= 0
  return y


def waveform_to_tacotron2_melspec(x):
  """Tacotron-2 style features: 24 kHz, nfft 1200, hop 300, -40 dB floor (spectral.py:230-247)."""
  return waveform_to_melspec(x, fs=24000, nfft=1200, nhop=300, norm_min_level_db=-40)


def waveform_to_r9y9_melspec(x, fs=22050):
  """r9y9/wavenet_vocoder features: nfft 1024, hop 256 (spectral.py:250-269)."""
  return waveform_to_melspec(x, fs=fs, nfft=1024, nhop=256)


def waveform_to_r9y9_melspec_tf(x, fs=22050):
  """Batched r9y9 features (spectral.py:272-291)."""
  return waveform_to_melspec_tf(x, fs=fs, nfft=1024, nhop=256)


# ---------------------------------------------------------------------------------------------
# inversion: the step after the generator (reference spectral.py:294-405)
# ---------------------------------------------------------------------------------------------
def _synthesis_window(nfft, nhop):
  """lws.synthwindow(awin, hop) = awin / sum_q awin^2[k + q*hop]; the denominator is 1 for the
  lws sqrt-Hann default at hop = nfft/4, so this equals the analysis window there."""
  def make():
    awin = _lws_window_f64(nfft, nhop)
    q = -(-nfft // nhop)
    sq = np.zeros(q * nhop, dtype=np.float64)
    sq[:nfft] = awin * awin
    den = np.tile(sq.reshape(q, nhop).sum(axis=0), q)[:nfft]
    return torch.from_numpy((awin / den).astype(np.float32))
  return _device_const(('swin', nfft, nhop), make)


def istft_batch(spec, nfft, nhop):
  """lws istft (perfectrec=False) of a batch: complex64 [clips, T, nfft//2+1] in HBM ->
  float32 [clips, (T-1)*nhop + nfft]."""
  _lib.require_device(spec)
  if spec.dtype != torch.complex64 or spec.dim() != 3 or spec.shape[2] != nfft // 2 + 1:
    raise ValueError('expected complex64 [clips, T, nfft//2+1]')
  spec = spec.contiguous()
  clips, T = spec.shape[0], spec.shape[1]
  n = (T - 1) * nhop + nfft if T > 0 else 0
  wav = torch.empty(clips, n, dtype=torch.float32, device=spec.device)
  if clips == 0 or T == 0:
    return wav
  work = torch.empty(clips, T, nfft, dtype=torch.float32, device=spec.device)
  _lib.check(_lib.load().advoc_istft_f32(
      _lib.ptr(torch.view_as_real(spec)), clips, T, _lib.ptr(_synthesis_window(nfft, nhop)),
      _lib.ptr(_device_twiddle(nfft)), nfft, nhop, _lib.ptr(work), _lib.ptr(wav), _lib.stream()),
      'advoc_istft_f32')
  return wav


def griffin_lim_batch(mag, nfft, nhop, ngl, unit_phase):
  """Griffin-Lim on a batch of equally long magnitude spectrograms, all on the GPU.
  mag, unit_phase: float32 [clips, T, bins] in HBM (unit_phase = the U[0,1) draw the reference
  takes from np.random.rand).  Returns float32 [clips, (T-1)*nhop + nfft].

  One iteration is three launches on static buffers (STFT; inverse FFT with the magnitude
  projection fused into its loads; overlap-add).  Replaying the iteration from a HIP graph was
  measured and bought nothing: at 32 clips the loop is bound by the kernels, not by launches."""
  lib = _lib.load()
  mag = mag.contiguous()
  unit_phase = unit_phase.contiguous()
  clips, T, bins = mag.shape
  dev = mag.device
  n = (T - 1) * nhop + nfft if T > 0 else 0
  wav = torch.empty(clips, n, dtype=torch.float32, device=dev)
  if clips == 0 or T == 0:
    return wav
  if nfft != 1024:
    raise _lib.AdvocHipError('Griffin-Lim runs on the 1024-point kernels only (nfft={})'.format(nfft))
  spec = torch.empty(clips, T, bins, 2, dtype=torch.float32, device=dev)
  work = torch.empty(clips, T, nfft, dtype=torch.float32, device=dev)
  awin, swin, tw = _device_window(nfft, nhop), _synthesis_window(nfft, nhop), _device_twiddle(nfft)

  def inverse():
    _lib.check(lib.advoc_istft_f32(_lib.ptr(spec), clips, T, _lib.ptr(swin), _lib.ptr(tw), nfft, nhop,
                                   _lib.ptr(work), _lib.ptr(wav), _lib.stream()), 'advoc_istft_f32')

  def iteration():
    # lws.stft of the (T-1)*hop + nfft samples gives T frames again (no padding); the projection
    # onto the given magnitudes is fused into the inverse transform's loads
    _lib.check(lib.advoc_stft_c64(_lib.ptr(wav), clips, n, _lib.ptr(awin), _lib.ptr(tw), nfft, nhop, T,
                                  _lib.ptr(spec), _lib.stream()), 'advoc_stft_c64')
    _lib.check(lib.advoc_istft_project_f32(_lib.ptr(spec), _lib.ptr(mag), clips, T, _lib.ptr(swin), _lib.ptr(tw),
                                           nfft, nhop, _lib.ptr(work), _lib.ptr(wav), _lib.stream()),
               'advoc_istft_project_f32')

  _lib.check(lib.advoc_polar_c64(_lib.ptr(mag), _lib.ptr(unit_phase), _lib.ptr(spec), mag.numel(),
                                 _lib.stream()), 'advoc_polar_c64')
  inverse()
  for _ in range(ngl):
    iteration()
  return wav


def magspec_to_waveform_griffin_lim(X_mag, nfft, nhop, ngl=60):
  """Reference spectral.py:294-311.  X_mag: nd-array [T, bins, 1] -> nd-array float32
  [(T-1)*nhop + nfft, 1, 1].  Initial phases come from numpy's global generator, exactly as in
  the reference (np.random.seed controls them); the iterations run in fp32 on the GPU (the
  reference iterates in float64 on the host)."""
  nsamps, nbins, nch = X_mag.shape
  if nch != 1:
    raise NotImplementedError('Can only invert monaural signals')
  X_mag = np.asarray(X_mag)[:, :, 0]
  u = np.random.rand(*X_mag.shape)
  mag = _to_device_f32(np.abs(X_mag).astype(np.float32))[None]
  wav = griffin_lim_batch(mag, nfft, nhop, ngl, _to_device_f32(u.astype(np.float32))[None])
  return wav[0].cpu().numpy()[:, np.newaxis, np.newaxis].astype(np.float32)


# LWS defaults (lws 1.2 `mode='speech'` as far as its documentation states them; restated, parity unpinned)
LWS_L = 5
LWS_LOOK_AHEAD = 3
LWS_NOFUTURE_THRESHOLDS = (1.0, 0.0)
LWS_ONLINE = (10, 1.0, 0.1)             # iterations, alpha, beta
LWS_BATCH = (100, 100.0, 0.1, 1.0)      # iterations, alpha, beta, gamma


def _lws_tables(nfft, nhop, L):
  """(weights [2Q-1, 2L-1, P] complex64 device constant, P): the STFT o iSTFT projection kernel alpha_q(p) times the
  frame rotation exp(-2 pi i r q nhop / nfft), r = (f + p) mod P, P = nfft / gcd(nfft, nhop); evaluated in float64."""
  P = nfft // math.gcd(nfft, nhop)

  def make():
    awin = _lws_window_f64(nfft, nhop)
    q_frames = -(-nfft // nhop)
    sq = np.zeros(q_frames * nhop, dtype=np.float64)
    sq[:nfft] = awin * awin
    swin = awin / np.tile(sq.reshape(q_frames, nhop).sum(axis=0), q_frames)[:nfft]
    n = np.arange(nfft)
    W = np.zeros((2 * q_frames - 1, 2 * L - 1, P), dtype=np.complex128)
    for q in range(-(q_frames - 1), q_frames):
      sh = np.zeros(nfft)
      lo, hi = max(0, q * nhop), min(nfft, nfft + q * nhop)
      if hi > lo:
        sh[lo:hi] = swin[lo - q * nhop:hi - q * nhop]
      prod = awin * sh
      rot = np.exp(-2j * np.pi * ((np.arange(P) * q * nhop) % nfft) / nfft)
      for p in range(-(L - 1), L):
        W[q + q_frames - 1, p + L - 1] = np.sum(prod * np.exp(2j * np.pi * p * n / nfft)) / nfft * rot
    return torch.view_as_real(torch.from_numpy(W.astype(np.complex64))).contiguous()
  return _device_const(('lwsW', nfft, nhop, L), make), P


def lws_spectrogram_batch(spec, nfft, nhop, L=LWS_L, look_ahead=LWS_LOOK_AHEAD,
                          nofuture_thresholds=LWS_NOFUTURE_THRESHOLDS, online=LWS_ONLINE, batch=LWS_BATCH):
  """Local Weighted Sums phase reconstruction of a batch of equally long spectrograms, all on the GPU
  (advoc_amd/csrc/lws.hip).  spec: float32 [clips, T, bins] magnitudes (phases start from nothing) or complex64
  [clips, T, bins] (its phases are the starting point, as lws.run_lws treats complex input).  Returns complex64
  [clips, T, bins]."""
  lib = _lib.load()
  _lib.require_device(spec)
  use_init = spec.is_complex()
  if use_init:
    c = torch.view_as_real(spec.to(torch.complex64)).contiguous()
    mag = torch.empty(c.shape[:-1], dtype=torch.float32, device=c.device)
    _lib.check(lib.advoc_cabs_f32(_lib.ptr(c), _lib.ptr(mag), mag.numel(), _lib.stream()), 'advoc_cabs_f32')
  else:
    mag = spec.abs().to(torch.float32).contiguous()       # (input sanitation, as lws.run_lws: magnitudes are |.|)
  clips, T, bins = mag.shape
  if bins != nfft // 2 + 1:
    raise ValueError('expected [clips, T, nfft//2+1]')
  dev = mag.device
  cur = torch.view_as_real(spec.to(torch.complex64)).contiguous().clone() if use_init else \
      torch.zeros(clips, T, bins, 2, dtype=torch.float32, device=dev)
  if clips == 0 or T == 0:
    return torch.view_as_complex(cur)
  W, P = _lws_tables(nfft, nhop, L)
  mean_mag = torch.empty(clips, dtype=torch.float32, device=dev)
  _lib.check(lib.advoc_lws_mean_mag_f32(_lib.ptr(mag), clips, T * bins, _lib.ptr(mean_mag), _lib.stream()),
             'advoc_lws_mean_mag_f32')
  thr = (ctypes.c_float * len(nofuture_thresholds))(*[float(v) for v in nofuture_thresholds])
  _lib.check(lib.advoc_lws_causal_c64(
      _lib.ptr(cur), _lib.ptr(mag), _lib.ptr(mean_mag), clips, T, nfft, nhop, _lib.ptr(W), P, L, look_ahead,
      thr, len(nofuture_thresholds), int(online[0]), float(online[1]), float(online[2]), int(use_init), _lib.stream()),
      'advoc_lws_causal_c64')
  n_sweeps = int(batch[0])
  if n_sweeps > 0:
    # all sweeps in one C call: sparse on the reference's (non-increasing) threshold schedule -- tiles with no bin above
    # the sweep's threshold are skipped (csrc/lws.hip)
    nxt = torch.empty_like(cur)
    ts = (ctypes.c_float * n_sweeps)(*[float(batch[1]) * math.exp(-float(batch[2]) * float(i) ** float(batch[3]))
                                        for i in range(n_sweeps)])
    tile_work = torch.empty(clips * ((T + 3) // 4), dtype=torch.float32, device=dev)      # >= clips * ceil(T / 8)
    _lib.check(lib.advoc_lws_batch_sweeps_c64(_lib.ptr(cur), _lib.ptr(nxt), _lib.ptr(mag), _lib.ptr(mean_mag), clips, T,
                                              nfft, nhop, _lib.ptr(W), P, L, ts, n_sweeps, _lib.ptr(tile_work),
                                              _lib.stream()), 'advoc_lws_batch_sweeps_c64')
  return torch.view_as_complex(cur)


def lws_batch(spec, nfft, nhop, **kw):
  """LWS phases + lws istft for a batch: [clips, T, bins] (float32 magnitudes or complex64) -> float32 waveforms
  [clips, (T-1)*nhop + nfft] in HBM."""
  if nfft != 1024:
    raise _lib.AdvocHipError('the inverse transform runs on the 1024-point kernels only (nfft={})'.format(nfft))
  return istft_batch(lws_spectrogram_batch(spec, nfft, nhop, **kw), nfft, nhop)


def magspec_to_waveform_lws(X_mag, nfft, nhop):
  """Reference spectral.py:314-326: lws.run_lws + lws.istft.  X_mag: nd-array [T, bins, 1], real magnitudes or --
  as the reference's own test passes it (tests/test_spectral.py:184,190) -- a complex spectrogram whose phases are the
  starting point.  Returns nd-array float32 [(T-1)*nhop + nfft, 1, 1].  The phase reconstruction is a restatement of the
  published LWS algorithm on the GPU in fp32 (the reference runs the third-party lws library in float64): same
  stages and defaults, not bit-comparable (parity unpinned)."""
  nsamps, nbins, nch = X_mag.shape
  if nch != 1:
    raise NotImplementedError('Can only invert monaural signals')
  X = np.asarray(X_mag)[:, :, 0]
  if np.iscomplexobj(X):
    spec = torch.from_numpy(X.astype(np.complex64)).to(_lib.device())[None]
  else:
    spec = _to_device_f32(np.abs(X).astype(np.float32))[None]
  wav = lws_batch(spec, nfft, nhop)
  return wav[0].cpu().numpy()[:, np.newaxis, np.newaxis].astype(np.float32)


def melspec_to_waveform(
    X_mel_dbnorm,
    fs,
    nfft,
    nhop,
    mel_min=125,
    mel_max=7600,
    norm_min_level_db=-100,
    norm_ref_level_db=20,
    phase_estimation='lws',
    waveform_len=None):
  """Approximately inverts a dB-normalised mel spectrogram to a waveform (reference
  spectral.py:330-395): de-normalise, pseudo-inverse mel basis, clamp at 0, phase estimation.

  Args:
    X_mel_dbnorm: nd-array float64 [?, mel_num_bins, 1].
    phase_estimation: 'lws' (the reference default; restated on the GPU) or 'gl<N>' (Griffin-Lim, N iterations).
    waveform_len: pad or clip the output to this length.
  Returns:
    nd-array float32 [waveform_len, 1, 1].
  """
  if X_mel_dbnorm.dtype != np.float64:
    raise ValueError()
  nsamps, mel_num_bins, nch = X_mel_dbnorm.shape
  if nch != 1:
    raise NotImplementedError('Can only invert monaural signals')
  X_mel_dbnorm = X_mel_dbnorm[:, :, 0]
  # host float64, as in the reference (:367-373): a [T,80]x[80,513] product, done once per utterance
  X_mel_db = (X_mel_dbnorm * -norm_min_level_db) + norm_min_level_db
  X_mel = np.power(10, (X_mel_db + norm_ref_level_db) / 20)
  inv_mel_filterbank = create_inverse_mel_filterbank(
      fs, nfft, fmin=mel_min, fmax=mel_max, n_mels=mel_num_bins)
  X_mag = np.dot(X_mel, inv_mel_filterbank.T)
  X_mag = np.maximum(0., X_mag)
  X_mag = X_mag[:, :, np.newaxis]
  if phase_estimation == 'lws':
    x = magspec_to_waveform_lws(X_mag, nfft, nhop)
  elif phase_estimation[:2] == 'gl':
    try:
      ngl = int(phase_estimation[2:])
    except Exception:
      raise ValueError()
    x = magspec_to_waveform_griffin_lim(X_mag, nfft, nhop, ngl)
  else:
    raise ValueError()
  if waveform_len is not None:
    x_len = x.shape[0]
    if x_len < waveform_len:
      x = np.pad(x, [[0, waveform_len - x_len], [0, 0], [0, 0]], 'constant')
    elif x_len > waveform_len:
      x = x[:waveform_len]
  return x.astype(np.float32)


def r9y9_melspec_to_waveform(X_mel_dbnorm, fs=22050, phase_estimation='lws', waveform_len=None):
  """Reference spectral.py:398-420."""
  return melspec_to_waveform(X_mel_dbnorm, fs=fs, nfft=1024, nhop=256,
                             phase_estimation=phase_estimation, waveform_len=waveform_len)
