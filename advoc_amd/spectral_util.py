"""Mel <-> linear-magnitude projections used around the model (drop-in for
/root/reference/models/advoc/spectral_util.py:6-60).

The filterbank [n_mels, 513] and its pseudo-inverse [513, n_mels] are float64 host constants
(advoc_amd.spectral) cast to float32 and kept in HBM; the projections run in
advoc_matmul_nt_f32.  As in the reference: linear-amplitude mel (no log), and NO >= 0 clamp
after the pseudo-inverse (spectral_util.py:34-43)."""
import numpy as np
import torch

from advoc_amd import _lib
from advoc_amd import spectral


class SpectralUtil(object):
  NFFT = 1024
  NHOP = 256
  FMIN = 125.
  FMAX = 7600.
  NMELS = 80
  fs = 22050
  FUSED_MAX_FRAMES = 65536      # largest call extract_training_triple hands to the one-launch extractor

  def __init__(self, n_mels=80, fs=22050):
    self.NMELS = n_mels
    self.fs = fs
    self.meltrans_np = spectral.create_mel_filterbank(
        self.fs, self.NFFT, fmin=self.FMIN, fmax=self.FMAX, n_mels=self.NMELS)
    self.invmeltrans_np = spectral.create_inverse_mel_filterbank(
        self.fs, self.NFFT, fmin=self.FMIN, fmax=self.FMAX, n_mels=self.NMELS)
    self._dev = {}

  def _const(self, name):
    dev = _lib.device()
    key = (name, dev.index)
    if key not in self._dev:
      if name == 'packed':
        self._dev[key] = spectral.pack_filterbank(self.meltrans_np, self.invmeltrans_np, dev)
      elif name == 'pairs':
        self._dev[key] = spectral.pack_inverse_pairs(self.invmeltrans_np, dev)
      else:
        src = self.meltrans_np if name == 'mel' else self.invmeltrans_np
        self._dev[key] = torch.from_numpy(src.astype(np.float32)).to(dev).contiguous()
    return self._dev[key]

  @property
  def meltrans(self):
    return self._const('mel')

  @property
  def invmeltrans(self):
    return self._const('inv')

  def extract_training_triple(self, wav):
    """What the train step consumes per batch (models/advoc/train_evaluate.py:55-56 on top of
    advoc/loader.py:116-128): waveforms [b, n, 1, 1] in HBM -> (|STFT| [b, T, 513, 1], linear mel [b, T, n_mels, 1],
    pseudo-inverted magnitudes [b, T, 513, 1]), T = 1 + (n - nfft) // nhop frames (whole frames only)."""
    wav = spectral._to_device_f32(wav)
    b, n, nfeats, ch = wav.shape
    T = 1 + (n - self.NFFT) // self.NHOP if n >= self.NFFT else 0
    if nfeats == 1 and ch == 1 and self.NMELS == 80 and self.NFFT == 1024 and 0 < b * T <= self.FUSED_MAX_FRAMES and \
        wav.dtype == torch.float32:
      # ONE launch: STFT, mel and pseudo-inverse without re-reading the magnitudes (csrc/extract.hip) -- faster than the
      # two launches up to ~64 k frames per call (the training feed is 2 x 64 clips x 256 frames = 32 k), slower above
      out = spectral.stft_mel_inverse(wav.reshape(b, n).contiguous(), self.NFFT, self.NHOP, T, self._const('packed'), self._const('pairs'))
      if out is not None:
        return out[0].unsqueeze(-1), out[1].unsqueeze(-1), out[2].unsqueeze(-1)
    mag = spectral.stft_magnitude(wav, self.NFFT, self.NHOP, pad_end=False)
    # both projections in one pass over the magnitudes (csrc/melpinv.hip)
    mel, inv = spectral.mel_and_inverse(mag[:, :, :, 0], self.meltrans, self.invmeltrans, packed=self._const('packed'))
    return mag, mel.unsqueeze(-1), inv.unsqueeze(-1)

  def mag_to_mel_linear_spec(self, mag_spec):
    """[B, T, 513, 1] -> [B, T, n_mels, 1]   (spectral_util.py:29-32)."""
    mag_spec = mag_spec.to(_lib.device(), torch.float32)
    return spectral.matmul_last(mag_spec[:, :, :, 0], self.meltrans).unsqueeze(-1)

  def mel_linear_to_mag_spec(self, mel_spec, transform='inverse'):
    """[B, T, n_mels, 1] -> [B, T, 513, 1]   (spectral_util.py:34-43)."""
    if transform != 'inverse':
      # the reference's 'transposed' branch reads an undefined name (spectral_util.py:38)
      raise NotImplementedError()
    mel_spec = mel_spec.to(_lib.device(), torch.float32)
    return spectral.matmul_last(mel_spec[:, :, :, 0], self.invmeltrans).unsqueeze(-1)

  def tacotron_mel_to_mag(self, X_mel_dbnorm):
    """dB-normalised mel [T, n_mels] (numpy float64) -> linear magnitude [T, 513], float32 tensor
    in HBM (spectral_util.py:52-60, scripts/spectrogram_advoc.py:15-22).  The de-normalisation is
    80 values per frame on the host in float64 as in the reference; the [T,80]x[80,513]
    projection runs on the GPU."""
    norm_min_level_db = -100
    norm_ref_level_db = 20
    X_mel_db = (np.asarray(X_mel_dbnorm, dtype=np.float64) * -norm_min_level_db) + norm_min_level_db
    X_mel = np.power(10, (X_mel_db + norm_ref_level_db) / 20)
    mel = torch.from_numpy(X_mel.astype(np.float32)).to(_lib.device())
    return spectral.matmul_last(mel, self.invmeltrans)

  def audio_from_mag_spec(self, mag_spec, phase_estimation='lws'):
    """Magnitude spectrogram [T, 513, 1] -> waveform float32 [n, 1, 1] (spectral_util.py:45-50: lws.run_lws + istft).
    'lws' runs the GPU restatement of LWS (advoc_amd.spectral.magspec_to_waveform_lws, parity unpinned); 'gl<N>'
    Griffin-Lim, the reference's own alternative (advoc/spectral.py:294-311)."""
    if phase_estimation == 'lws':
      return spectral.magspec_to_waveform_lws(np.asarray(mag_spec), self.NFFT, self.NHOP)
    if phase_estimation[:2] != 'gl':
      raise ValueError()
    return spectral.magspec_to_waveform_griffin_lim(np.asarray(mag_spec, dtype=np.float64), self.NFFT, self.NHOP,
                                                    int(phase_estimation[2:]))
