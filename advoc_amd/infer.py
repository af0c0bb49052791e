"""Mel spectrogram -> magnitude spectrogram with the AdVoc generator (the GPU part of the
reference's vocoding script, /root/reference/scripts/spectrogram_advoc.py:15-22, 80-94).

The reference feeds 256-frame chunks to the generator ONE AT A TIME through `sess.run`
(:88-92); chunks are independent (no overlap-add), so here they go through the generator as
batches.  Kept exactly: de-normalisation + pseudo-inverse without a >= 0 clamp (:15-22), padding to
``int(T / subseq_len) * subseq_len + subseq_len`` frames -- i.e. one extra all-zero chunk when T
is already a multiple (:83-84) --, trimming back to T (:93-94), dropout active at inference.
Phase estimation (LWS, :95) is the next row of the build and is not performed here.
"""
import numpy as np
import torch

from advoc_amd import _lib
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
