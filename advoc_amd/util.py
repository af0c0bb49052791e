"""Small helpers of the reference's advoc/util.py and models/melspecgan/util.py.

The reference versions are TF graph ops feeding tf.summary (advoc/util.py:36-62,
models/melspecgan/util.py:7-37); here they take and return numpy arrays or
torch tensors.  ``*_to_approx_audio`` defaults to LWS like the reference (the GPU restatement in
advoc_amd.spectral).
"""
import numpy as np
import torch

from . import spectral


def best_shape(t, axis=None):
  """advoc/util.py:7-33.  Shapes are always static here."""
  dims = [int(d) for d in t.shape]
  return dims if axis is None else dims[axis]


def r9y9_melspec_norm(x):
  return (x * 2.) - 1.


def r9y9_melspec_denorm(x):
  return (x + 1.) * 0.5


def r9y9_melspec_to_uint8_img(x):
  """[B, T, F, 1] in [0, 1] -> uint8 image batch [B, F, T, 1], frequency upwards
  (tf.image.rot90: counter-clockwise quarter turn of each image)."""
  if isinstance(x, torch.Tensor):
    img = torch.rot90(x, 1, dims=(1, 2))
    return (img * 255.).clamp(0., 255.).to(torch.uint8)
  img = np.rot90(np.asarray(x), 1, axes=(1, 2))
  return np.clip(img * 255., 0., 255.).astype(np.uint8)


def r9y9_melspec_to_approx_audio(x, fs, waveform_len, n=None, phase_estimation='lws'):
  """[B, T, 80, 1] dB-normalised mel -> float32 waveforms [B', waveform_len, 1, 1]."""
  if n is not None:
    x = x[:n]
  if isinstance(x, torch.Tensor):
    x = x.detach().cpu().numpy()
  out = [spectral.r9y9_melspec_to_waveform(np.asarray(item, dtype=np.float64), fs=fs,
                                           phase_estimation=phase_estimation,
                                           waveform_len=waveform_len) for item in x]
  return np.stack(out).astype(np.float32)


feats_norm = r9y9_melspec_norm
feats_denorm = r9y9_melspec_denorm
feats_to_uint8_img = r9y9_melspec_to_uint8_img
feats_to_approx_audio = r9y9_melspec_to_approx_audio
