"""advoc_amd.util: the summary helpers of the reference's advoc/util.py:36-62 and
models/melspecgan/util.py:7-37 (host code; the audio helper needs the GPU)."""
import numpy as np
import pytest
import torch

from advoc_amd import util


def test_norm_roundtrip_and_image():
  rng = np.random.RandomState(0)
  x = rng.rand(2, 6, 4, 1).astype(np.float32)
  assert np.allclose(util.r9y9_melspec_denorm(util.r9y9_melspec_norm(x)), x, atol=1e-7)
  assert np.allclose(util.feats_denorm(util.feats_norm(x)), x, atol=1e-7)
  img = util.r9y9_melspec_to_uint8_img(x * 1.2 - 0.1)     # exercises both clip ends
  assert img.dtype == np.uint8 and img.shape == (2, 4, 6, 1)
  # counter-clockwise quarter turn: the highest mel bin becomes the top row
  want = np.clip((x * 1.2 - 0.1) * 255., 0., 255.).astype(np.uint8)
  assert np.array_equal(img[:, 0, :, 0], want[:, :, -1, 0])
  assert np.array_equal(img[:, -1, :, 0], want[:, :, 0, 0])
  timg = util.feats_to_uint8_img(torch.from_numpy(x * 1.2 - 0.1))
  assert np.array_equal(timg.numpy(), img)
  assert util.best_shape(torch.zeros(3, 5, 7)) == [3, 5, 7]
  assert util.best_shape(torch.zeros(3, 5, 7), axis=1) == 5


@pytest.mark.gpu
def test_approx_audio_shapes():
  rng = np.random.RandomState(1)
  x = rng.rand(4, 64, 80, 1).astype(np.float32)
  wav = util.feats_to_approx_audio(x, 16000, 16384, n=3)
  assert wav.shape == (3, 16384, 1, 1) and wav.dtype == np.float32
  assert np.isfinite(wav).all() and float(np.abs(wav).max()) > 0.
