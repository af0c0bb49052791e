"""torch-CPU restatement of the MelspecGAN generator's INFERENCE graph.

TEST INFRASTRUCTURE (see oracle/__init__.py).  STRUCTURE PINNED (r3) by the one TensorFlow-written artefact the
reference holds, models/melspecgan/infer.meta: tests/test_melspecgan_graph.py evaluates that graph op by op
(tests/tf_graph_interp.py, Conv2DBackpropInput literally as the autograd gradient of a SAME conv) and requires
`generator` below to equal it to float64 round-off, and the variable table / hyper-parameters to match.  No trained
weights or golden tensors are reachable, so there is no numeric known answer.

Restated (paths relative to /root/reference):
  models/melspecgan/conv2d.py:4-14     dense_layer            x @ W + b, W [in, out]
  models/melspecgan/conv2d.py:17-53    conv2d_transpose_layer tf.nn.conv2d_transpose(k=5, s=2, 'SAME'),
                                       W [kh, kw, out_ch, in_ch]: output = 2 x input; the implicit padding of
                                       the matching SAME conv (total k - s = 3) puts 1 before, 2 after
  models/melspecgan/conv2d.py:82-150   MelspecGANGenerator.__call__(z, training=False)
  models/melspecgan/train.py:156-165   infer graph: G_z = feats_denorm(G(z))
  models/melspecgan/util.py:11-12      feats_denorm = (x + 1) / 2
  tf.layers.batch_normalization(training=False): (x - moving_mean) * gamma / sqrt(moving_variance + 1e-3) + beta
"""
import torch
import torch.nn.functional as F

Z_DIM = 100
BN_EPS = 1e-3          # tf.layers.batch_normalization default epsilon


def variable_specs(dim=64, kernel_len=5, batchnorm=True):
  """[(TF variable name under scope 'G', shape)] in creation order."""
  specs = [('G/z_proj/W', (Z_DIM, 4 * 5 * dim * 8)), ('G/z_proj/b', (4 * 5 * dim * 8,))]
  chans = [dim * 8, dim * 4, dim * 2, dim, 1]

  def bn(i, c):
    base = 'G/batch_normalization' + ('' if i == 0 else '_%d' % i)
    return [(base + '/gamma', (c,)), (base + '/beta', (c,)), (base + '/moving_mean', (c,)),
            (base + '/moving_variance', (c,))]
  if batchnorm:
    specs += bn(0, chans[0])
  for i in range(4):
    specs.append(('G/upconv_%d/W' % (i + 1), (kernel_len, kernel_len, chans[i + 1], chans[i])))
    specs.append(('G/upconv_%d/b' % (i + 1), (chans[i + 1],)))
    if batchnorm and i < 3:
      specs += bn(i + 1, chans[i + 1])
  return specs


def init_params(dim=64, seed=0, batchnorm=True, dtype=torch.float32):
  """N(0, 0.02) weights, zero biases (conv2d.py:7-11,41-51); BN statistics randomised so the
  inference-mode affine is exercised (a freshly initialised TF graph would have mean 0 / variance 1)."""
  g = torch.Generator().manual_seed(seed)
  P = {}
  for name, shape in variable_specs(dim, 5, batchnorm):
    if name.endswith('/W'):
      P[name] = (torch.randn(shape, generator=g) * 0.02).to(dtype)
    elif name.endswith('/b') or name.endswith('/beta'):
      P[name] = (torch.randn(shape, generator=g) * 0.05).to(dtype)
    elif name.endswith('/gamma'):
      P[name] = (1.0 + 0.1 * torch.randn(shape, generator=g)).to(dtype)
    elif name.endswith('/moving_mean'):
      P[name] = (0.1 * torch.randn(shape, generator=g)).to(dtype)
    else:
      P[name] = (0.5 + torch.rand(shape, generator=g)).to(dtype)
  return P


def conv2d_transpose_same(x, W, b, stride=2):
  """x [B,H,W,Cin], W [kh,kw,Cout,Cin] -> [B, stride*H, stride*W, Cout]."""
  full = F.conv_transpose2d(x.permute(0, 3, 1, 2), W.permute(3, 2, 0, 1).contiguous(), None, stride=stride)
  oh, ow = stride * x.shape[1], stride * x.shape[2]
  y = full[:, :, 1:1 + oh, 1:1 + ow].permute(0, 2, 3, 1)
  return y + b


def batchnorm_infer(x, P, i):
  base = 'G/batch_normalization' + ('' if i == 0 else '_%d' % i)
  return (x - P[base + '/moving_mean']) * (P[base + '/gamma'] * torch.rsqrt(P[base + '/moving_variance'] + BN_EPS)) \
      + P[base + '/beta']


def generator(P, z, dim=64, batchnorm=True, denorm=True):
  """z [B,100] -> [B,64,80,1]: tanh output, mapped to [0,1] by feats_denorm when `denorm`."""
  x = z @ P['G/z_proj/W'] + P['G/z_proj/b']
  x = x.reshape(-1, 4, 5, dim * 8)
  for i in range(4):
    if i < 4:
      if batchnorm:
        x = batchnorm_infer(x, P, i)
      x = torch.relu(x)
    x = conv2d_transpose_same(x, P['G/upconv_%d/W' % (i + 1)], P['G/upconv_%d/b' % (i + 1)])
  x = torch.tanh(x)
  return (x + 1.) * 0.5 if denorm else x
