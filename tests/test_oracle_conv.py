"""Closed-form checks of the conv-stack oracle itself (oracle/advoc_torch.py), on the CPU.

The reference holds no test or golden tensor for the networks (PARITY UNPINNED, NOTEBOOK.md §2), so
the oracle's transcription of the TF1 semantics is pinned here against hand-derivable cases: the
asymmetric SAME padding, the transposed conv as the adjoint of the SAME conv, the tie gradient of
tf.maximum, inference of the (1,2)-stride rule, the losses on a two-element example and TF's Adam
update written out."""
import math

import numpy as np
import pytest
import torch

from oracle import advoc_torch as A


def test_same_pad_puts_the_extra_row_after():
  assert A.same_pad(513, 4, 2) == (1, 2)      # odd width: total 3 -> 1 before, 2 after
  assert A.same_pad(256, 4, 2) == (1, 1)
  assert A.same_pad(1, 4, 1) == (1, 2)        # the collapsed time axis, stride 1
  assert A.same_pad(5, 4, 1) == (1, 2)


def test_gen_conv_counts_in_bounds_taps():
  """All-ones 4x4 kernel on an all-ones 4x5 image, stride 2: each output = number of taps that land
  inside the image under (1,1) x (1,2) padding."""
  x = torch.ones(1, 4, 5, 1, dtype=torch.float64)
  w = torch.ones(4, 4, 1, 1, dtype=torch.float64)
  y = A.gen_conv(x, w, torch.zeros(1, dtype=torch.float64))[0, :, :, 0]
  rows = [sum(1 for k in range(4) if 0 <= 2 * o - 1 + k < 4) for o in range(2)]      # [3, 3]
  cols = [sum(1 for k in range(4) if 0 <= 2 * o - 1 + k < 5) for o in range(3)]      # [3, 4, 2]
  assert y.shape == (2, 3)
  assert torch.equal(y, torch.tensor([[r * c for c in cols] for r in rows], dtype=torch.float64))
  assert cols == [3, 4, 2]                                                            # the asymmetry


@pytest.mark.parametrize('strides,hw', [((2, 2), (6, 9)), ((1, 2), (1, 9)), ((1, 2), (2, 5))])
def test_gen_deconv_is_the_adjoint_of_the_same_conv(strides, hw):
  """tf.layers.conv2d_transpose('same') is defined as the input-gradient of the SAME conv whose
  output has the transposed conv's input shape: <deconv(x), u> == <x, conv(u)>."""
  g = torch.Generator().manual_seed(0)
  H, W = hw
  x = torch.randn(2, H, W, 3, generator=g, dtype=torch.float64)
  w = torch.randn(4, 4, 5, 3, generator=g, dtype=torch.float64)          # [kh, kw, out, in]
  y = A.gen_deconv(x, w, torch.zeros(5, dtype=torch.float64), strides=strides)
  assert y.shape == (2, strides[0] * H, strides[1] * W, 5)
  u = torch.randn(y.shape, generator=g, dtype=torch.float64)
  # forward SAME conv from the deconv's OUTPUT space to its INPUT space uses kernel [kh, kw, in=5, out=3]
  conv = A.gen_conv(u, w, torch.zeros(3, dtype=torch.float64), strides)
  assert conv.shape == x.shape
  assert abs(float((y * u).sum() - (conv * x).sum())) < 1e-9 * float(y.norm() * u.norm())


def test_discrim_conv_pads_one_on_every_side():
  x = torch.ones(1, 5, 6, 1, dtype=torch.float64)
  w = torch.ones(4, 4, 1, 1, dtype=torch.float64)
  y = A.discrim_conv(x, w, torch.zeros(1, dtype=torch.float64), 1)[0, :, :, 0]
  assert y.shape == (4, 5) and float(y[0, 0]) == 9 and float(y[1, 1]) == 16 and float(y[3, 4]) == 9


def test_lrelu_value_and_tie_gradient():
  x = torch.tensor([-2.0, 0.0, 3.0], requires_grad=True)
  y = A.lrelu(x)
  assert torch.allclose(y, torch.tensor([-0.4, 0.0, 3.0]))
  y.sum().backward()
  assert torch.allclose(x.grad, torch.tensor([0.2, 0.2, 1.0]))    # TF MaximumGrad: the tie goes to alpha*x


def test_batchnorm_is_training_mode_with_eps_1e_5():
  g = torch.Generator().manual_seed(1)
  x = torch.randn(3, 4, 5, 2, generator=g, dtype=torch.float64)
  y = A.batchnorm(x, torch.tensor([2.0, 1.0], dtype=torch.float64), torch.tensor([0.5, -1.0], dtype=torch.float64))
  m, v = x.mean(dim=(0, 1, 2)), x.var(dim=(0, 1, 2), unbiased=False)
  assert torch.allclose(y, (x - m) / torch.sqrt(v + 1e-5) * torch.tensor([2.0, 1.0]) + torch.tensor([0.5, -1.0]))


def test_encoder_stride_rule():
  s = lambda small, n: [a for a, _ in A.encoder_strides(A.Config(small=small, subseq_len=n))]   # noqa: E731
  assert s(False, 256) == [2] * 8
  assert s(False, 64) == [2, 2, 2, 2, 2, 2, 1, 1]          # models/advoc/melspecVocoder.py:114
  assert s(True, 256) == [2] * 5 and s(True, 16) == [2, 2, 2, 2, 1]


def test_losses_on_a_hand_example():
  """advoc_model.py:238-245 with EPS = 1e-12 written out for two patch probabilities."""
  p_real, p_fake = torch.tensor([0.8, 0.6]), torch.tensor([0.3, 0.1])
  d = torch.mean(-(torch.log(p_real + A.EPS) + torch.log(1 - p_fake + A.EPS)))
  want_d = -0.5 * (math.log(0.8) + math.log(0.7) + math.log(0.6) + math.log(0.9))
  assert abs(float(d) - want_d) < 1e-6
  assert abs(float(torch.mean(-torch.log(p_fake + A.EPS))) - (-0.5 * (math.log(0.3) + math.log(0.1)))) < 1e-6


def test_adam_is_tensorflows_formulation():
  """tf.train.AdamOptimizer: lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); p -= lr_t m / (sqrt(v) + eps) --
  epsilon OUTSIDE the bias correction (unlike torch.optim.Adam)."""
  P = {'w': torch.tensor([1.0, -2.0], dtype=torch.float64)}
  opt = A.AdamTF(['w'], P)
  g1, g2 = torch.tensor([0.5, -1e-9], dtype=torch.float64), torch.tensor([-0.25, 2e-9], dtype=torch.float64)
  opt.step(P, {'w': g1})
  lr1 = 0.0002 * math.sqrt(1 - 0.999) / (1 - 0.5)
  m1, v1 = 0.5 * g1, 0.001 * g1 * g1
  want = torch.tensor([1.0, -2.0], dtype=torch.float64) - lr1 * m1 / (torch.sqrt(v1) + 1e-8)
  assert torch.allclose(P['w'], want, rtol=0, atol=1e-15)
  # a gradient of 1e-9: without epsilon the step would be the full ~lr * 15.8 = 2e-4; epsilon damps it 300x
  assert abs(float(want[1] + 2.0)) < 1e-6
  opt.step(P, {'w': g2})
  lr2 = 0.0002 * math.sqrt(1 - 0.999 ** 2) / (1 - 0.5 ** 2)
  m2, v2 = 0.5 * m1 + 0.5 * g2, 0.999 * v1 + 0.001 * g2 * g2
  assert torch.allclose(P['w'], want - lr2 * m2 / (torch.sqrt(v2) + 1e-8), rtol=0, atol=1e-15)


def test_dropout_is_inverted_dropout():
  x = torch.tensor([2.0, 4.0, 6.0])
  assert torch.equal(A.dropout(x, torch.tensor([1.0, 0.0, 1.0]), 0.5), torch.tensor([4.0, 0.0, 12.0]))


def test_gen_deconv_is_tensorflow_conv2d_backprop_input():
  """tf.layers.conv2d_transpose(k=4, strides, 'same') runs TF's Conv2DBackpropInput.  tests/tf_graph_interp.py implements
  that op from its definition (the autograd gradient of a SAME-padded strided conv) and is validated on the
  TensorFlow-written MelspecGAN graph (tests/test_melspecgan_graph.py); the AdVoc oracle's transposed conv
  (models/advoc/advoc_model.py:53-69) must be the same function, for both stride settings the model uses."""
  import tf_graph_interp as interp
  g = torch.Generator().manual_seed(0)
  for strides, (h, w) in (((2, 2), (5, 9)), ((1, 2), (1, 5)), ((2, 2), (1, 3))):
    x = torch.randn(2, h, w, 6, generator=g, dtype=torch.float64)
    kern = torch.randn(4, 4, 3, 6, generator=g, dtype=torch.float64)          # [kh, kw, out, in]
    got = A.gen_deconv(x, kern, torch.zeros(3, dtype=torch.float64), strides=strides)
    want = interp.conv2d_backprop_input([2, strides[0] * h, strides[1] * w, 3], kern, x, [1, strides[0], strides[1], 1],
                                        'SAME', 'NHWC')
    assert tuple(got.shape) == tuple(want.shape)
    assert float((got - want).abs().max()) < 1e-12
