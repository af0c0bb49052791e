"""torch-CPU restatement of the reference's AdVoc generator / discriminator / losses / Adam.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: the reference
holds no test, golden tensor or checkpoint for this part of the path
(SURVEY.md §8c "Unpinned"); TF1 is not installable here.  This file transcribes
the TF1 graph semantics op by op and is itself cross-checked against
closed-form micro cases and a pure-python loop conv in tests/test_oracle_conv.py.

Restated (paths relative to /root/reference):
  models/advoc/advoc_model.py:25-32    _discrim_conv  (pad 1, VALID, k=4)
  models/advoc/advoc_model.py:34-51    _gen_conv      (tf.layers.conv2d SAME, k=4)
  models/advoc/advoc_model.py:53-69    _gen_deconv    (tf.layers.conv2d_transpose SAME, k=4)
  models/advoc/advoc_model.py:75-166   build_generator
  models/advoc/advoc_model.py:168-204  build_discriminator
  models/advoc/advoc_model.py:238-257  losses, var split, Adam(2e-4, 0.5) x2
  models/advoc/advoc_model.py:285-289  train_loop (D on batch k, G on batch k+1)
  models/advoc/advoc_model_small.py:14-15,22,107-108,128-129,134  small variant

TF1 semantics transcribed (third party, not in /root/reference):
  * SAME padding: out=ceil(in/s), pad_total=max((out-1)*s+k-in,0), before=pad_total//2
  * conv2d kernel [kh,kw,in,out]; conv2d_transpose kernel [kh,kw,out,in], output = in*s
  * tf.nn.dropout: (x / keep) * floor(keep + u)           (mask injected here)
  * batch_normalization(training=True): batch mean / biased var, eps 1e-5
  * AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t*m/(sqrt(v)+eps)

Tensors are NHWC ([B, time, freq, C]) like the reference; parameters are kept in
TF layouts under TF variable names so the dict doubles as the checkpoint map.
"""
import collections
import math

import torch
import torch.nn.functional as F

EPS = 1e-12


# ----------------------------------------------------------------------------
# configuration (advoc_model.py:10-22 / advoc_model_small.py:10-23)
# ----------------------------------------------------------------------------
class Config(object):
  def __init__(self, small=False, **kw):
    self.audio_fs = 22050
    self.subseq_len = 256
    self.n_mels = 80
    self.ngf = 32 if small else 64
    self.ndf = 32 if small else 64
    self.gan_weight = 1.
    self.l1_weight = 10.
    self.use_batchnorm = False
    self.small = small
    self.num_enc_layers = 4 if small else 7
    self.nbins = 513
    for k, v in kw.items():
      setattr(self, k, v)

  def encoder_channels(self):
    g = self.ngf
    return [g] + [g * 2, g * 4, g * 8, g * 8, g * 8, g * 8, g * 8][:self.num_enc_layers]

  def decoder_specs(self):
    """[(name_index, out_channels, dropout)] for decoder_N .. decoder_2 (decoder_1 separate)."""
    g = self.ngf
    if self.small:
      full = [(g * 8, .5), (g * 8, .5), (g * 8, .5), (g * 8, .5), (g * 4, .5), (g * 2, 0.), (g, 0.)]
      full = full[len(full) - self.num_enc_layers:]
    else:
      full = [(g * 8, .5), (g * 8, .5), (g * 8, .5), (g * 8, 0.), (g * 4, 0.), (g * 2, 0.), (g, 0.)]
    n_enc = 1 + self.num_enc_layers
    return [(n_enc - i, c, d) for i, (c, d) in enumerate(full)]


def same_pad(n, k, s):
  out = -(-n // s)
  tot = max((out - 1) * s + k - n, 0)
  return tot // 2, tot - tot // 2


# ----------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------
def init_params(cfg, seed=0, dtype=torch.float32):
  """N(0, 0.02) kernels, zero biases, BN gamma=1 beta=0 (advoc_model.py:30,36,55)."""
  gen = torch.Generator().manual_seed(seed)
  P = collections.OrderedDict()

  def kern(name, shape):
    P[name] = (torch.randn(shape, generator=gen, dtype=torch.float32) * 0.02).to(dtype)

  def bias(name, n):
    P[name] = torch.zeros(n, dtype=dtype)

  def bn(scope, n):
    if cfg.use_batchnorm:
      P[scope + '/batch_normalization/gamma'] = torch.ones(n, dtype=dtype)
      P[scope + '/batch_normalization/beta'] = torch.zeros(n, dtype=dtype)

  enc = cfg.encoder_channels()
  cin = 1
  for i, c in enumerate(enc):
    s = 'generator/encoder_%d' % (i + 1)
    kern(s + '/conv2d/kernel', (4, 4, cin, c))
    bias(s + '/conv2d/bias', c)
    if i > 0:
      bn(s, c)
    cin = c
  prev = enc[-1]
  for j, (idx, c, _) in enumerate(cfg.decoder_specs()):
    s = 'generator/decoder_%d' % idx
    cin = prev if j == 0 else prev + enc[idx - 1]
    kern(s + '/conv2d_transpose/kernel', (4, 4, c, cin))
    bias(s + '/conv2d_transpose/bias', c)
    bn(s, c)
    prev = c
  s = 'generator/decoder_1'
  kern(s + '/conv2d_transpose/kernel', (4, 4, 1, prev + enc[0]))
  bias(s + '/conv2d_transpose/bias', 1)

  d = cfg.ndf
  chans = [d, d * 2, d * 4, d * 8, 1]
  cin = 2
  for i, c in enumerate(chans):
    s = 'discriminator/layer_%d' % (i + 1)
    kern(s + '/conv2d/kernel', (4, 4, cin, c))
    bias(s + '/conv2d/bias', c)
    if 1 <= i <= 3:
      bn(s, c)
    cin = c
  return P


def split_vars(P):
  G = [k for k in P if k.startswith('generator')]
  D = [k for k in P if k.startswith('discriminator')]
  return G, D


# ----------------------------------------------------------------------------
# ops (NHWC in / out)
# ----------------------------------------------------------------------------
def _nchw(x):
  return x.permute(0, 3, 1, 2)


def _nhwc(x):
  return x.permute(0, 2, 3, 1)


def lrelu(x, alpha=0.2):
  # tf.maximum(alpha * x, x) (advoc_model.py:86-87); written with where() so that the gradient at
  # x == 0 is alpha, as TF's MaximumGrad gives (tie -> first argument), not torch.maximum's 0.6
  return torch.where(x > 0, x, alpha * x)


def gen_conv(x, kernel, bias, strides=(2, 2)):
  pt, pb = same_pad(x.shape[1], 4, strides[0])
  pl, pr = same_pad(x.shape[2], 4, strides[1])
  xp = F.pad(_nchw(x), (pl, pr, pt, pb))
  return _nhwc(F.conv2d(xp, kernel.permute(3, 2, 0, 1).contiguous(), bias, stride=strides))


def gen_deconv(x, kernel, bias, strides=(2, 2)):
  # tf.layers.conv2d_transpose(k=4, 'same') is the input-gradient of a SAME conv: output = stride x
  # input, and the (k - s) implicit padding puts 1 row/column BEFORE (the smaller share, as SAME
  # does) for s = 2 and for s = 1 alike.  Full transposed conv, then crop [1, 1 + s*n).
  full = F.conv_transpose2d(_nchw(x), kernel.permute(3, 2, 0, 1).contiguous(), bias, stride=strides, padding=0)
  oh, ow = strides[0] * x.shape[1], strides[1] * x.shape[2]
  return _nhwc(full[:, :, 1:1 + oh, 1:1 + ow])


def encoder_strides(cfg):
  """advoc_model.py:90-116: encoder_1 always strides (2,2); later encoders stride (2,2) while the
  running n_time (a float: `n_time /= 2`) is > 1, then (1,2).  Returns the list for encoder_1..N."""
  n_time = cfg.subseq_len / 2
  out = [(2, 2)]
  for _ in range(cfg.num_enc_layers):
    if n_time > 1:
      out.append((2, 2))
      n_time /= 2
    else:
      out.append((1, 2))
  return out


def discrim_conv(x, kernel, bias, stride):
  xp = F.pad(_nchw(x), (1, 1, 1, 1))
  return _nhwc(F.conv2d(xp, kernel.permute(3, 2, 0, 1).contiguous(), bias, stride=stride))


def batchnorm(x, gamma, beta, eps=1e-5):
  mean = x.mean(dim=(0, 1, 2), keepdim=True)
  var = ((x - mean) ** 2).mean(dim=(0, 1, 2), keepdim=True)
  return (x - mean) * torch.rsqrt(var + eps) * gamma + beta


def dropout(x, mask, keep):
  return (x / keep) * mask


# ----------------------------------------------------------------------------
# networks
# ----------------------------------------------------------------------------
def dropout_shapes(cfg, batch):
  """{'decoder_N': shape} of every dropout mask the generator consumes."""
  shapes = {}
  enc = cfg.encoder_channels()
  h, w = cfg.subseq_len, cfg.nbins
  es = encoder_strides(cfg)
  n_stride1 = sum(1 for st in es if st == (1, 2))
  for sh, sw in es:
    h, w = -(-h // sh), -(-w // sw)
  for j, (idx, c, drop) in enumerate(cfg.decoder_specs()):
    if j > 0:
      w -= 1
    h, w = h * (1 if j < n_stride1 else 2), w * 2
    if drop > 0:
      shapes['decoder_%d' % idx] = (batch, h, w, c)
  return shapes


def make_dropout_masks(cfg, batch, seed, dtype=torch.float32):
  gen = torch.Generator().manual_seed(seed)
  return {k: (torch.rand(s, generator=gen) >= 0.5).to(dtype)
          for k, s in dropout_shapes(cfg, batch).items()}


def _gated(t, gate, slope):
  """slope = 0.2: lrelu, 0: relu -- with the GATE (x > 0) given from outside instead of taken from t: the activation a
  consumer applies when the sign pattern is frozen (tests: a comparison that cannot be decided by which side of zero a
  round-off error puts a pre-activation value; gate None = the ordinary activation)."""
  if gate is None:
    return lrelu(t, slope) if slope else torch.relu(t)
  return torch.where(gate, t, slope * t)


def build_generator(P, x, cfg, masks, collect=None, gates=None):
  """gates (optional, tests only): {layer name: bool tensor shaped like that layer's output} -- the sign pattern every
  consumer of that output uses in its (leaky) ReLU instead of the output's own signs."""
  gates = gates or {}
  es = encoder_strides(cfg)
  n_stride1 = sum(1 for st in es if st == (1, 2))
  bnorm = (lambda t, s: batchnorm(t, P[s + '/batch_normalization/gamma'],
                                  P[s + '/batch_normalization/beta'])) if cfg.use_batchnorm else (lambda t, s: t)
  layers, names = [], []

  def gate_of(k):          # of layers[k], or None
    return gates.get(names[k])

  def cat_gate(a, b):      # gate of cat([layers[a] trimmed, layers[b]]): both given or neither
    ga, gb = gate_of(a), gate_of(b)
    if ga is None and gb is None:
      return None
    ga = ga if ga is not None else layers[a] > 0
    gb = gb if gb is not None else layers[b] > 0
    return torch.cat([ga[:, :, :-1, :], gb], dim=3)
  s = 'generator/encoder_1'
  layers.append(gen_conv(x, P[s + '/conv2d/kernel'], P[s + '/conv2d/bias']))
  names.append('encoder_1')
  for i in range(1, 1 + cfg.num_enc_layers):
    s = 'generator/encoder_%d' % (i + 1)
    out = gen_conv(_gated(layers[-1], gate_of(-1), 0.2), P[s + '/conv2d/kernel'], P[s + '/conv2d/bias'], strides=es[i])
    layers.append(bnorm(out, s))
    names.append('encoder_%d' % (i + 1))
  for j, (idx, c, drop) in enumerate(cfg.decoder_specs()):
    s = 'generator/decoder_%d' % idx
    inp = layers[-1] if j == 0 else torch.cat([layers[-1][:, :, :-1, :], layers[idx - 1]], dim=3)
    g_in = gate_of(-1) if j == 0 else cat_gate(len(layers) - 1, idx - 1)
    out = gen_deconv(_gated(inp, g_in, 0.0), P[s + '/conv2d_transpose/kernel'], P[s + '/conv2d_transpose/bias'],
                     strides=(1, 2) if j < n_stride1 else (2, 2))       # advoc_model.py:139-142
    out = bnorm(out, s)
    if drop > 0:
      out = dropout(out, masks['decoder_%d' % idx], 1 - drop)
    layers.append(out)
    names.append('decoder_%d' % idx)
  s = 'generator/decoder_1'
  inp = torch.cat([layers[-1][:, :, :-1, :], layers[0]], dim=3)
  out = gen_deconv(_gated(inp, cat_gate(len(layers) - 1, 0), 0.0), P[s + '/conv2d_transpose/kernel'],
                   P[s + '/conv2d_transpose/bias'])
  out = out[:, :, :-1, :]
  layers.append(out)
  if collect is not None:
    collect.extend(layers)
  return out


def build_discriminator(P, cond, target, cfg, collect=None, gates=None, tag=''):
  """gates (optional, tests only): {tag + 'layer_k': bool tensor} -- the sign pattern of layer k's leaky ReLU, k = 1 .. 4
  (see build_generator)."""
  gates = gates or {}
  bnorm = (lambda t, s: batchnorm(t, P[s + '/batch_normalization/gamma'],
                                  P[s + '/batch_normalization/beta'])) if cfg.use_batchnorm else (lambda t, s: t)
  x = torch.cat([cond, target], dim=3)
  s = 'discriminator/layer_1'
  h = _gated(discrim_conv(x, P[s + '/conv2d/kernel'], P[s + '/conv2d/bias'], 2), gates.get(tag + 'layer_1'), 0.2)
  acts = [h]
  for i in range(3):
    s = 'discriminator/layer_%d' % (i + 2)
    stride = 1 if i == 2 else 2
    h = _gated(bnorm(discrim_conv(h, P[s + '/conv2d/kernel'], P[s + '/conv2d/bias'], stride), s),
               gates.get(tag + 'layer_%d' % (i + 2)), 0.2)
    acts.append(h)
  s = 'discriminator/layer_5'
  out = torch.sigmoid(discrim_conv(h, P[s + '/conv2d/kernel'], P[s + '/conv2d/bias'], 1))
  acts.append(out)
  if collect is not None:
    collect.extend(acts)
  return out


def losses(P, x, target, cfg, masks, gates=None):
  """advoc_model.py:217-245 -> dict of scalars + gen output."""
  gen = build_generator(P, x, cfg, masks, gates=gates)
  p_real = build_discriminator(P, x, target, cfg, gates=gates, tag='D/real/')
  p_fake = build_discriminator(P, x, gen, cfg, gates=gates, tag='D/fake/')
  d_loss = torch.mean(-(torch.log(p_real + EPS) + torch.log(1 - p_fake + EPS)))
  g_gan = torch.mean(-torch.log(p_fake + EPS))
  g_l1 = torch.mean(torch.abs(target - gen))
  if cfg.gan_weight > 0:
    g_loss = g_gan * cfg.gan_weight + g_l1 * cfg.l1_weight
  else:
    g_loss = g_l1 * cfg.l1_weight
  return dict(gen=gen, p_real=p_real, p_fake=p_fake, d_loss=d_loss, g_gan=g_gan, g_l1=g_l1,
              g_loss=g_loss)


def grads(P, x, target, cfg, masks, which, gates=None):
  """d(loss)/d(vars): which='D' -> discrim_loss wrt D vars; 'G' -> gen_loss wrt G vars.  gates: build_generator."""
  Gk, Dk = split_vars(P)
  keys = Dk if which == 'D' else Gk
  Q = collections.OrderedDict((k, v.detach().clone().requires_grad_(k in keys)) for k, v in P.items())
  L = losses(Q, x, target, cfg, masks, gates=gates)
  loss = L['d_loss'] if which == 'D' else L['g_loss']
  g = torch.autograd.grad(loss, [Q[k] for k in keys])
  return collections.OrderedDict(zip(keys, g)), {k: v.detach() for k, v in L.items()}


def grads_in_chunks(P, x, target, cfg, masks, which, chunk, gates=None):
  """grads() evaluated `chunk` clips at a time (bounded memory at BASELINE batch sizes).  Valid WITHOUT batch norm only:
  every loss of advoc_model.py:238-245 is a mean over the batch of per-clip terms, so the batch gradient is the
  clip-count-weighted mean of the chunk gradients and the losses are the weighted means of the chunk losses."""
  assert not cfg.use_batchnorm
  B = x.shape[0]
  total, info = None, {}
  for lo in range(0, B, chunk):
    hi = min(B, lo + chunk)
    wgt = (hi - lo) / float(B)
    g, L = grads(P, x[lo:hi], target[lo:hi], cfg, {k: v[lo:hi] for k, v in masks.items()}, which,
                 gates={k: v[lo:hi] for k, v in gates.items()} if gates else None)
    if total is None:
      total = collections.OrderedDict((k, v * wgt) for k, v in g.items())
    else:
      for k, v in g.items():
        total[k] += v * wgt
    for k in ('d_loss', 'g_gan', 'g_l1', 'g_loss'):
      info[k] = info.get(k, 0.0) + float(L[k]) * wgt
  return total, info


# ----------------------------------------------------------------------------
# TF AdamOptimizer(0.0002, 0.5) (advoc_model.py:250-257)
# ----------------------------------------------------------------------------
class AdamTF(object):
  def __init__(self, keys, P, lr=0.0002, beta1=0.5, beta2=0.999, eps=1e-8):
    self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
    self.t = 0
    self.m = {k: torch.zeros_like(P[k]) for k in keys}
    self.v = {k: torch.zeros_like(P[k]) for k in keys}

  def step(self, P, G):
    self.t += 1
    lr_t = self.lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
    for k, g in G.items():
      self.m[k] = self.b1 * self.m[k] + (1 - self.b1) * g
      self.v[k] = self.b2 * self.v[k] + (1 - self.b2) * g * g
      P[k] = P[k] - lr_t * self.m[k] / (torch.sqrt(self.v[k]) + self.eps)


class Trainer(object):
  """advoc_model.py:285-289: D update on one batch, then G update on the next."""

  def __init__(self, cfg, seed=0, dtype=torch.float32):
    self.cfg = cfg
    self.P = init_params(cfg, seed, dtype)
    Gk, Dk = split_vars(self.P)
    self.g_opt = AdamTF(Gk, self.P)
    self.d_opt = AdamTF(Dk, self.P)
    self.step = 0

  def train_loop(self, batch_d, batch_g, masks_d, masks_g):
    info = {}
    if self.cfg.gan_weight > 0:
      x, target = batch_d
      g, L = grads(self.P, x, target, self.cfg, masks_d, 'D')
      self.d_opt.step(self.P, g)
      info['d_loss'] = float(L['d_loss'])
    x, target = batch_g
    g, L = grads(self.P, x, target, self.cfg, masks_g, 'G')
    self.g_opt.step(self.P, g)
    self.step += 1
    info.update(g_loss=float(L['g_loss']), g_gan=float(L['g_gan']), g_l1=float(L['g_l1']))
    return self.step, info
