"""MelspecGAN generator, inference only: z ~ N(0, I)^100 -> a 64-frame, 80-bin dB-normalised mel
spectrogram in [0, 1] (drop-in for the INFERENCE graph of /root/reference/models/melspecgan:
conv2d.py:82-150 `MelspecGANGenerator.__call__(z, training=False)`, train.py:156-165 `infer`,
scripts/generate_spectrogram.py:37-56).  Training (WGAN-GP, train.py:95-111) needs double backward
and stays out of scope.

The five layers run on the same HIP kernels as the AdVoc generator:
  z_proj     dense 100 -> 4*5*8d       advoc_matmul_nt_f32 (bias folded in as an extra input column)
  upconv_1-3 5x5 stride-2 transposed   advoc_conv_forward (sub-pixel phases on the fp32 MFMA gather GEMM)
  upconv_4   5x5, one output channel   advoc_conv_forward (direct kernel)
  tanh + feats_denorm                  advoc_tanh_affine_f32
Inference-mode batch norm is a per-channel affine; it and the ReLU that follows are applied by the
CONSUMING layer as it loads its input (in_scale / in_shift / ACT_RELU), so the normalised tensors are
never written.  Parameters carry the TF variable names (`G/z_proj/W`, `G/upconv_1/W`,
`G/batch_normalization_2/moving_mean`, ...), so a TF checkpoint loads by name (advoc_amd.tf_checkpoint).
"""
import collections

import numpy as np
import torch

from advoc_amd import _lib
from advoc_amd import conv as C

Z_DIM = 100
BN_EPS = 1e-3     # tf.layers.batch_normalization default


def feats_norm(x):
  return (x * 2.) - 1.


def feats_denorm(x):
  return (x + 1.) * 0.5


class MelspecGANGenerator(object):
  def __init__(self, dim=64, kernel_len=5, batchnorm=True):
    if kernel_len != 5:
      raise NotImplementedError('kernel_len {}: the reference only instantiates 5'.format(kernel_len))
    if dim % 32:
      raise NotImplementedError('dim must be a multiple of 32 (MFMA channel tiling)')
    self.dim = dim
    self.kernel_len = kernel_len
    self.stride = 2
    self.batchnorm = batchnorm
    self._built = None
    self._params = None

  # ---- parameters ----
  def variable_specs(self):
    d = self.dim
    chans = [d * 8, d * 4, d * 2, d, 1]
    specs = [('G/z_proj/W', (Z_DIM, 4 * 5 * d * 8)), ('G/z_proj/b', (4 * 5 * d * 8,))]

    def bn(i, c):
      base = 'G/batch_normalization' + ('' if i == 0 else '_%d' % i)
      return [(base + '/gamma', (c,)), (base + '/beta', (c,)), (base + '/moving_mean', (c,)),
              (base + '/moving_variance', (c,))]
    if self.batchnorm:
      specs += bn(0, chans[0])
    for i in range(4):
      specs.append(('G/upconv_%d/W' % (i + 1), (5, 5, chans[i + 1], chans[i])))
      specs.append(('G/upconv_%d/b' % (i + 1), (chans[i + 1],)))
      if self.batchnorm and i < 3:
        specs += bn(i + 1, chans[i + 1])
    return specs

  def _ensure_params(self, seed=0):
    if self._params is not None:
      return
    _lib.load()
    dev = _lib.device()
    g = torch.Generator().manual_seed(seed)
    P = collections.OrderedDict()
    for name, shape in self.variable_specs():
      if name.endswith('/W'):
        t = torch.randn(shape, generator=g) * 0.02            # conv2d.py:7-8,41-42
      elif name.endswith('/gamma') or name.endswith('/moving_variance'):
        t = torch.ones(shape)
      else:
        t = torch.zeros(shape)
      P[name] = t.to(dev)
    self._params = P
    self._derived = None

  def state_dict(self):
    self._ensure_params()
    return collections.OrderedDict((k, v.clone()) for k, v in self._params.items())

  def load_state_dict(self, state):
    self._ensure_params()
    for k, v in self._params.items():
      if k not in state:
        raise KeyError('missing variable {!r}'.format(k))
      t = torch.as_tensor(np.asarray(state[k]) if not isinstance(state[k], torch.Tensor) else state[k])
      if tuple(t.shape) != tuple(v.shape):
        raise ValueError('{}: shape {} != {}'.format(k, tuple(t.shape), tuple(v.shape)))
      v.copy_(t.to(v.device, torch.float32))
    self._derived = None

  def load_tf_checkpoint(self, prefix):
    from advoc_amd import tf_checkpoint
    want = [n for n, _ in self.variable_specs()]
    found = tf_checkpoint.read_checkpoint(prefix, names=lambda n: n in set(want) or n == 'global_step')
    missing = [n for n in want if n not in found]
    if missing:
      raise KeyError('TF checkpoint {!r} lacks {}'.format(prefix, missing[:4]))
    self.load_state_dict({k: torch.from_numpy(np.array(found[k], dtype=np.float32)) for k in want})
    return int(found['global_step']) if 'global_step' in found else 0

  # ---- graph ----
  def _derive(self):
    """Tensors computed once per parameter set: the dense kernel as [out, in + 1] with the bias as the
    last input column, and the inference-mode BN affines (host-side float64, 5 vectors)."""
    if self._derived is not None:
      return self._derived
    P = self._params
    W = torch.cat([P['G/z_proj/W'], P['G/z_proj/b'][None, :]], dim=0)      # [101, out]
    aff = []
    for i in range(4):
      if not self.batchnorm:
        aff.append((None, None))
        continue
      base = 'G/batch_normalization' + ('' if i == 0 else '_%d' % i)
      g, b = P[base + '/gamma'].double().cpu(), P[base + '/beta'].double().cpu()
      m, v = P[base + '/moving_mean'].double().cpu(), P[base + '/moving_variance'].double().cpu()
      sc = g / torch.sqrt(v + BN_EPS)
      aff.append((sc.float().to(W.device), (b - m * sc).float().to(W.device)))
    self._derived = dict(Wt=W.t().contiguous(), aff=aff)
    self._built = None
    return self._derived

  def build(self, batch_size):
    self._ensure_params()
    D = self._derive()
    if self._built is not None and self._built['B'] == batch_size:
      return self
    dev = self._params['G/z_proj/W'].device
    f32 = dict(dtype=torch.float32, device=dev)
    d = self.dim
    chans = [d * 8, d * 4, d * 2, d, 1]
    B = int(batch_size)
    acts = [torch.zeros(B, 4, 5, chans[0], **f32)]
    h, w = 4, 5
    layers = []
    for i in range(4):
      h, w = 2 * h, 2 * w
      acts.append(torch.zeros(B, h, w, chans[i + 1], **f32))
      sc, sh = D['aff'][i]
      layers.append(C.Layer(C.DECONV, acts[i], acts[i + 1], self._params['G/upconv_%d/W' % (i + 1)],
                            self._params['G/upconv_%d/b' % (i + 1)], stride=(2, 2), pad=(1, 1),
                            in_act=C.ACT_RELU, in_scale=sc, in_shift=sh))
    self._built = dict(B=B, acts=acts, layers=layers, zin=torch.ones(B, Z_DIM + 1, **f32),
                       out=torch.zeros(B, 64, 80, 1, **f32))
    return self

  def __call__(self, z, training=False, denorm=False):
    """z: [B, 100] float32 (numpy or torch) -> torch tensor [B, 64, 80, 1] in HBM: tanh output in
    [-1, 1], or feats_denorm of it (what the reference's `G_z` tensor holds) when `denorm`."""
    if training:
      raise NotImplementedError('MelspecGAN training (WGAN-GP) is outside this build; inference only')
    z = torch.as_tensor(z)
    if z.dim() != 2 or z.shape[1] != Z_DIM:
      raise ValueError('z must be [batch, {}]'.format(Z_DIM))
    self.build(z.shape[0])
    st, lib = self._built, _lib.load()
    st['zin'][:, :Z_DIM].copy_(z.to(st['zin'].device, torch.float32))
    Wt = self._derived['Wt']
    a0 = st['acts'][0]
    _lib.check(lib.advoc_matmul_nt_f32(_lib.ptr(st['zin']), _lib.ptr(Wt), _lib.ptr(a0), st['B'], Z_DIM + 1,
                                       Wt.shape[0], _lib.stream()), 'advoc_matmul_nt_f32')
    for lay in st['layers']:
      lay.forward()
    last = st['acts'][-1]
    _lib.check(lib.advoc_tanh_affine_f32(_lib.ptr(last), _lib.ptr(st['out']), last.numel(),
                                         0.5 if denorm else 1.0, 0.5 if denorm else 0.0, _lib.stream()),
               'advoc_tanh_affine_f32')
    return st['out'].clone()

  def generate(self, n, batch_size=64, seed=None):
    """n samples of G_z = feats_denorm(G(z)), z ~ N(0, I) (train.py:157-163): numpy float32 [n, 64, 80, 1]."""
    g = torch.Generator().manual_seed(seed) if seed is not None else None
    out = []
    for lo in range(0, n, batch_size):
      b = min(batch_size, n - lo)
      z = torch.randn(b, Z_DIM, generator=g)
      out.append(self(z, denorm=True).cpu().numpy())
    return np.concatenate(out, axis=0) if out else np.zeros((0, 64, 80, 1), np.float32)
