"""Mel spectrogram -> magnitude spectrogram with the AdVoc generator (the GPU part of the
reference's vocoding script, /root/reference/scripts/spectrogram_advoc.py:15-22, 80-94).

The reference feeds 256-frame chunks to the generator ONE AT A TIME through `sess.run`
(:88-92); chunks are independent (no overlap-add), so here they go through the generator as
batches.  Kept exactly: de-normalisation + pseudo-inverse without a >= 0 clamp (:15-22), padding to
``int(T / subseq_len) * subseq_len + subseq_len`` frames -- i.e. one extra all-zero chunk when T
is already a multiple (:83-84) --, trimming back to T (:93-94), dropout active at inference.
Phase estimation: the reference uses LWS (:95); `vocode_batch` / the script run Griffin-Lim on the GPU.
"""
import numpy as np
import torch

from advoc_amd.spectral_util import SpectralUtil


def chunk_plan(nframes, subseq_len):
  """(padded length, number of chunks) as spectrogram_advoc.py:83-85 computes them."""
  target = int(nframes / subseq_len) * subseq_len + subseq_len
  return target, int(target / subseq_len)


def vocode_melspec(model, spec, spectral_util=None, chunk_batch=16):
  """spec: nd-array float64 [T, n_mels, 1] (dB-normalised mel, scripts/audio_to_spectrogram.py)
  -> generated magnitude spectrogram, nd-array float32 [T, 513, 1]."""
  su = spectral_util or SpectralUtil(n_mels=model.n_mels, fs=model.audio_fs)
  subseq_len = model.subseq_len
  X_mag = su.tacotron_mel_to_mag(np.asarray(spec)[:, :, 0])            # [T, 513] float32 in HBM
  T = X_mag.shape[0]
  target, n = chunk_plan(T, subseq_len)
  padded = torch.zeros(target, X_mag.shape[1], dtype=torch.float32, device=X_mag.device)
  padded[:T] = X_mag
  chunks = padded.reshape(n, subseq_len, X_mag.shape[1], 1)
  outs = []
  for lo in range(0, n, chunk_batch):
    outs.append(model.build_generator(chunks[lo:lo + chunk_batch]))
  gen = torch.cat(outs, dim=0).reshape(target, X_mag.shape[1], 1)[:T]
  return gen.cpu().numpy()


def vocode_batch(model, specs, spectral_util=None, phase_estimation='gl60', chunk_batch=64, unit_phase=None):
  """Equally long mel spectrograms -> waveforms, all on the GPU: nd-array / tensor float [n, T, n_mels, 1]
  (dB-normalised, e.g. MelspecGAN samples) -> (magnitudes float32 [n, T, 513] in HBM, waveforms float32
  [n, (T-1)*256 + 1024] in HBM or None).  Per sample exactly what scripts/spectrogram_advoc.py:80-95 /
  models/advoc/melspecVocoder.py:57-83 do (pseudo-inverse, padding to int(T/L)*L + L frames -- a whole
  extra zero chunk when T is a multiple of L --, generator, trim), with every chunk of every sample
  in one generator batch and phase reconstruction over all samples at once.  phase_estimation: 'lws' (what the
  reference script runs, scripts/spectrogram_advoc.py:95), 'gl<N>' or None."""
  from advoc_amd import spectral
  su = spectral_util or SpectralUtil(n_mels=model.n_mels, fs=model.audio_fs)
  specs = torch.as_tensor(np.asarray(specs) if not isinstance(specs, torch.Tensor) else specs)
  n, T = specs.shape[0], specs.shape[1]
  L = model.subseq_len
  mel_host = specs[:, :, :, 0].reshape(n * T, -1).detach().cpu().numpy()
  X_mag = su.tacotron_mel_to_mag(mel_host).reshape(n, T, -1)                     # [n, T, 513] in HBM
  target, per = chunk_plan(T, L)
  padded = torch.zeros(n, target, X_mag.shape[2], dtype=torch.float32, device=X_mag.device)
  padded[:, :T] = X_mag
  chunks = padded.reshape(n * per, L, X_mag.shape[2], 1)
  outs = []
  for lo in range(0, n * per, chunk_batch):
    outs.append(model.build_generator(chunks[lo:lo + chunk_batch]))
  gen = torch.cat(outs, dim=0).reshape(n, target, X_mag.shape[2])[:, :T].contiguous()
  if phase_estimation is None:
    return gen, None
  if phase_estimation == 'lws':
    return gen, spectral.lws_batch(gen.abs().contiguous(), SpectralUtil.NFFT, SpectralUtil.NHOP)
  if phase_estimation[:2] != 'gl':
    raise ValueError("phase_estimation {!r}: expected 'lws', 'gl<iterations>' (e.g. 'gl60') or None".format(phase_estimation))
  if unit_phase is None:
    unit_phase = torch.rand(gen.shape, device=gen.device)
  wav = spectral.griffin_lim_batch(gen.abs(), SpectralUtil.NFFT, SpectralUtil.NHOP, int(phase_estimation[2:]),
                                   unit_phase)
  return gen, wav


def load_generator(ckpt_fp, model_type='regular', subseq_len=256, fs=22050):
  """Builds an INFER-mode model and restores generator weights from a train_evaluate.py checkpoint."""
  from advoc_amd.model import Advoc, AdvocSmall, Modes
  from advoc_amd.train_evaluate import restore_checkpoint
  model = (AdvocSmall if model_type == 'small' else Advoc)(Modes.INFER)
  model.subseq_len = subseq_len
  model.audio_fs = fs
  model.build(batch_size=1)
  restore_checkpoint(ckpt_fp, model, with_optimizer=False)
  return model
