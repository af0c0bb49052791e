"""MelspecGAN generator inference on MI355X vs the torch-CPU oracle and vs the TensorFlow-written inference graph of the
reference (models/melspecgan/infer.meta decoded into tests/golden/melspecgan_graph.json; the graph's STRUCTURE is pinned by
it, its trained weights are not reachable)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import melspecgan_torch as M

gpu = pytest.mark.gpu


def rel(a, b):
  a = torch.as_tensor(a).double().cpu()
  b = torch.as_tensor(b).double().cpu()
  return float((a - b).norm() / b.norm().clamp_min(1e-30))


def test_oracle_shapes_and_variable_names():
  P = M.init_params(dim=32, seed=1)
  names = [n for n, _ in M.variable_specs(32)]
  assert names[:2] == ['G/z_proj/W', 'G/z_proj/b'] and 'G/batch_normalization_3/moving_variance' in names
  assert len(names) == 2 + 4 * 4 + 4 * 2                      # conv2d.py asserts 8 BN update ops = 4 layers
  y = M.generator(P, torch.randn(2, 100), dim=32)
  assert tuple(y.shape) == (2, 64, 80, 1) and float(y.min()) >= 0 and float(y.max()) <= 1
  # the 5x5 stride-2 SAME transposed conv is the adjoint of the 5x5 stride-2 SAME conv (pad 1 before, 2 after)
  x = torch.randn(1, 4, 5, 3, dtype=torch.float64)
  W = torch.randn(5, 5, 2, 3, dtype=torch.float64)
  y = M.conv2d_transpose_same(x, W, torch.zeros(2, dtype=torch.float64))
  u = torch.randn_like(y)
  up = torch.nn.functional.pad(u.permute(0, 3, 1, 2), (1, 2, 1, 2))
  conv = torch.nn.functional.conv2d(up, W.permute(3, 2, 0, 1), stride=2).permute(0, 2, 3, 1)
  assert abs(float((y * u).sum() - (conv * x).sum())) < 1e-9 * float(y.norm() * u.norm())


@gpu
@pytest.mark.parametrize('shape', [(2, 4, 5, 256, 128), (3, 8, 10, 64, 32), (2, 16, 20, 32, 1), (1, 3, 7, 64, 64)])
def test_transposed_conv_5x5_forward(hip, shape):
  from advoc_amd import _lib, conv
  B, H, W, cin, cout = shape
  g = torch.Generator().manual_seed(H)
  x = torch.randn(B, H, W, cin, generator=g)
  w = torch.randn(5, 5, cout, cin, generator=g) * 0.05
  b = torch.randn(cout, generator=g) * 0.1
  sc = 1 + 0.1 * torch.randn(cin, generator=g)
  sh = 0.1 * torch.randn(cin, generator=g)
  want = M.conv2d_transpose_same(torch.relu(x.double() * sc.double() + sh.double()), w.double(), b.double())
  dev = torch.device('cuda')
  y = torch.full((B, 2 * H, 2 * W, cout), float('nan'), device=dev)
  L = conv.Layer(conv.DECONV, x.to(dev), y, w.to(dev), b.to(dev), stride=(2, 2), pad=(1, 1), in_act=conv.ACT_RELU,
                 in_scale=sc.to(dev), in_shift=sh.to(dev))
  L.forward()
  assert torch.isfinite(y).all() and rel(y, want) < 2e-5
  # inference only: the backward directions of 5x5 kernels are refused, not mis-computed
  with pytest.raises(_lib.AdvocHipError, match='unsupported'):
    L.backward_data(torch.zeros_like(y), torch.zeros(B, H, W, cin, device=dev))
  with pytest.raises(_lib.AdvocHipError, match='unsupported'):
    L.backward_weight(torch.zeros_like(y), torch.zeros(5, 5, cout, cin, device=dev))


@gpu
@pytest.mark.parametrize('dim,batchnorm', [(32, True), (64, True), (32, False)])
def test_generator_matches_oracle(hip, dim, batchnorm):
  from advoc_amd.melspecgan import MelspecGANGenerator
  P = M.init_params(dim=dim, seed=3, batchnorm=batchnorm)
  G = MelspecGANGenerator(dim=dim, batchnorm=batchnorm)
  G.load_state_dict(P)
  z = torch.randn(5, 100, generator=torch.Generator().manual_seed(9))
  P64 = {k: v.double() for k, v in P.items()}
  for denorm in (False, True):
    want = M.generator(P64, z.double(), dim=dim, batchnorm=batchnorm, denorm=denorm)
    got = G(z, denorm=denorm)
    assert tuple(got.shape) == (5, 64, 80, 1) and got.dtype == torch.float32
    assert rel(got, want) < 1e-4, rel(got, want)
  assert tuple(G(z[:2]).shape) == (2, 64, 80, 1)              # another batch size rebinds
  with pytest.raises(NotImplementedError):
    G(z, training=True)
  with pytest.raises(ValueError):
    G(torch.zeros(3, 99))


@gpu
def test_generator_matches_the_tensorflow_graph(hip):
  """The HIP generator (dim 64, batch norm: the configuration the reference exported) against the TF-written graph
  evaluated op by op in float64 (tests/tf_graph_interp.py) -- no oracle transcription in between."""
  import json
  import tf_graph_interp as interp
  from advoc_amd.melspecgan import MelspecGANGenerator
  with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'melspecgan_graph.json')) as f:
    graph = json.load(f)
  P = M.init_params(dim=64, seed=11)
  G = MelspecGANGenerator(dim=64, batchnorm=True)
  G.load_state_dict(P)
  z = torch.randn(4, 100, generator=torch.Generator().manual_seed(5))
  want = interp.run(graph, 'G_z', {'z': z.double()}, {k: v.double() for k, v in P.items()})
  got = G(z, denorm=True)
  assert tuple(got.shape) == tuple(want.shape) == (4, 64, 80, 1)
  assert rel(got, want) < 1e-4, rel(got, want)
  pre = interp.run(graph, 'G/Tanh', {'z': z.double()}, {k: v.double() for k, v in P.items()})
  assert rel(G(z, denorm=False), pre) < 1e-4


@gpu
def test_generate_script_and_tf_checkpoint(hip, tmp_path):
  """scripts/generate_spectrogram.py: TF-container checkpoint in, zero-padded .npy files out
  (reference :49-56); the samples feed scripts/spectrogram_advoc.py (heuristic mode) to audio."""
  from advoc_amd import tf_checkpoint
  from advoc_amd.melspecgan import MelspecGANGenerator
  P = M.init_params(dim=32, seed=4)
  tensors = {k: v.numpy() for k, v in P.items()}
  tensors['global_step'] = np.array(4321, dtype=np.int64)
  prefix = str(tmp_path / 'model.ckpt-4321')
  tf_checkpoint.write_checkpoint(prefix, tensors)
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  out = tmp_path / 'mels'
  r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'generate_spectrogram.py'), '--out_dir', str(out),
                      '--ckpt_fp', prefix, '--n', '5', '--b', '2', '--dim', '32', '--seed', '7'],
                     capture_output=True, text=True)
  assert r.returncode == 0, r.stderr[-2000:]
  assert 'Restored from step 4321' in r.stdout
  files = sorted(os.listdir(out))
  assert files == ['%09d.npy' % i for i in range(5)]
  G = MelspecGANGenerator(dim=32)
  assert G.load_tf_checkpoint(prefix) == 4321
  z = torch.randn(2, 100, generator=torch.Generator().manual_seed(7))
  first = np.load(out / files[0])
  assert first.shape == (64, 80, 1) and first.dtype == np.float32
  assert np.abs(first - G(z, denorm=True)[0].cpu().numpy()).max() < 1e-6
  # z -> mel -> (pseudo-inverse heuristic) -> Griffin-Lim waveform: the joint pipeline of BASELINE configs[4]
  wavs = tmp_path / 'wavs'
  r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'spectrogram_advoc.py'), '--spec_dir', str(out),
                      '--out_dir', str(wavs), '--fs', '16000', '--phase_estimation', 'gl3'],
                     capture_output=True, text=True)
  assert r.returncode == 0, r.stderr[-2000:]
  assert sorted(os.listdir(wavs)) == ['%09d.wav' % i for i in range(5)]
