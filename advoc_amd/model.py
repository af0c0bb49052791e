"""AdVoc generator / discriminator / train step on MI355X.

Drop-in for the model objects of /root/reference/models/advoc:
  model.py:1-18            Model, Modes
  util.py:1-20             override_model_attrs
  advoc_model.py:9-289     class Advoc        (full: ngf = ndf = 64, 8 encoders)
  advoc_model_small.py     class Advoc(small) (ngf = ndf = 32, 1 + 4 encoders)  -> AdvocSmall here

What is different by construction (MI355X-first, not a TF1 graph):
  * no graph / session: layers are bound once to pre-allocated HBM buffers (a static plan of
    C-ABI calls, advoc_amd/conv.py) and replayed every step;
  * parameters, gradients and Adam slots of each network live in ONE flat fp32 arena
    (laid out in backward-completion order) so the optimiser is a single kernel launch and
    data-parallel gradient exchange is a few large RCCL all-reduces;
  * real and fake discriminator passes of the D step run as one 2B batch;
  * every elementwise op of the reference graph (lrelu / relu, concat, [:, :, :-1, :] trims, bias,
    dropout, sigmoid + log losses) is fused into the conv / loss kernels.
Semantics kept: TF SAME padding, kernel layouts and variable names, N(0, 0.02) init,
dropout 0.5 in every mode, D update on one batch then G update on the NEXT batch, TF Adam.
"""
import collections
import contextlib
import math
import os

import torch

from advoc_amd import _lib
from advoc_amd import conv as C

EPS = 1e-12


class Modes(object):
  TRAIN = 'train'
  EVAL = 'eval'
  INFER = 'infer'


class Model(object):
  def __init__(self, mode, *args, **kwargs):
    self.mode = mode

  def __call__(self):
    raise Exception('Abstract method')

  def train_loop(self):
    raise Exception('Abstract method')

  def eval_ckpt(self, ckpt_fp):
    raise Exception('Abstract method')


def override_model_attrs(model, overrides):
  """'a=b,c=d' -> setattr with the type of the current attribute (reference util.py:1-20).
  Returns (model, 'attr,value' summary of public non-callable attributes, sorted)."""
  if overrides is not None and len(overrides.strip()):
    for pair in overrides.split(','):
      key, val = pair.split('=')
      kind = type(getattr(model, key))
      if kind == bool:
        val = val in ['True', 'true', 't', '1']
      elif kind == list:
        val = val.split(';')
      else:
        val = kind(val)
      setattr(model, key, val)
  names = sorted(k for k in dir(model) if not k.startswith('_') and not callable(getattr(model, k)))
  return model, '\n'.join('{},{}'.format(k, getattr(model, k)) for k in names)


class _Arena(object):
  """Flat fp32 storage for a set of named tensors; every entry starts 16-byte aligned."""

  def __init__(self, specs, device):
    self.offsets = collections.OrderedDict()
    off = 0
    for name, shape in specs:
      self.offsets[name] = (off, tuple(shape))
      n = 1
      for s in shape:
        n *= s
      off += (n + 3) // 4 * 4
    self.size = off
    self.device = device

  def new(self):
    return torch.zeros(self.size, dtype=torch.float32, device=self.device)

  def views(self, flat):
    out = collections.OrderedDict()
    for name, (off, shape) in self.offsets.items():
      n = 1
      for s in shape:
        n *= s
      out[name] = flat[off:off + n].view(shape)
    return out


class Advoc(Model):
  audio_fs = 22050
  subseq_len = 256
  n_mels = 80
  ngf = 64
  ndf = 64
  gan_weight = 1.
  l1_weight = 10.
  train_batch_size = 8
  eval_batch_size = 1
  separable_conv = False
  use_batchnorm = False
  generator_type = "pix2pix"  # only "pix2pix" is on the MI355X hot path

  # --- structure (advoc_model.py:96-128) ---
  _enc_mult = [2, 4, 8, 8, 8, 8, 8]
  _dec_spec = [(8, 0.5), (8, 0.5), (8, 0.5), (8, 0.0), (4, 0.0), (2, 0.0), (1, 0.0)]
  _nbins = 513

  # Adam (advoc_model.py:250-251)
  _lr, _beta1, _beta2, _adam_eps = 0.0002, 0.5, 0.999, 1e-8

  def __init__(self, mode, *args, **kwargs):
    super(Advoc, self).__init__(mode)
    self._built = None
    self._seed = 0
    self._feed = None
    self._injected_masks = None
    self._dropout_calls = 0
    # public attributes appear in the printed attr summary (util.py:13-18); like the reference,
    # `step`, `G_vars` and `D_vars` only exist once the model has been built
    self._world_size = 1
    self._rank = 0
    self._allreduce = None
    self._reduce_async = None      # (start(flat, lo, hi), finish(), bucket_elems) from parallel.DataParallel
    self._sync_bn = None           # in-place cross-replica sum of a small float64 tensor (batch-norm statistics)

  # ------------------------------------------------------------------------------------------
  # structure helpers
  # ------------------------------------------------------------------------------------------
  def _encoder_channels(self):
    return [self.ngf] + [self.ngf * m for m in self._enc_mult]

  def _decoder_specs(self):
    """[(decoder index, out channels, dropout)] for decoder_N .. decoder_2."""
    n_enc = len(self._encoder_channels())
    return [(n_enc - i, self.ngf * m, d) for i, (m, d) in enumerate(self._dec_spec)]

  def _check_supported(self):
    if self.generator_type != 'pix2pix':
      raise NotImplementedError('generator_type {!r}: only "pix2pix" runs on the MI355X path'
                                .format(self.generator_type))
    if self.separable_conv:
      raise NotImplementedError('separable_conv=True is a non-default ablation outside the hot path')
    if self.ngf % 32 or self.ndf % 32:
      raise NotImplementedError('ngf / ndf must be multiples of 32: the gfx950 MFMA kernels tile channels '
                                'by 32 (reference defaults: 64, small model: 32)')
    # the skip concats only line up when halving (SAME: ceil) and doubling retrace each other;
    # the reference's TF graph fails to build otherwise (tf.concat shape error)
    h, hs = int(self.subseq_len), []
    for sh, _ in self._encoder_strides():
      h = -(-h // sh)
      hs.append(h)
    hd = hs[-1]
    n1 = sum(1 for st in self._encoder_strides() if st == (1, 2))
    for j in range(len(hs) - 1):
      hd *= 1 if j < n1 else 2
      if hd != hs[len(hs) - 2 - j]:
        raise ValueError('subseq_len {}: decoder heights do not match the encoder skips'.format(self.subseq_len))
    if 2 * hs[0] != int(self.subseq_len):
      raise ValueError('subseq_len must be even')
    # PatchGAN: three stride-2 convs, then two 4x4 stride-1 convs with pad 1 (each -1): needs T/8 - 2 >= 1
    if self.mode == Modes.TRAIN and int(self.subseq_len) // 8 - 2 < 1:
      raise ValueError('subseq_len {} leaves the discriminator no output rows (needs >= 24; TF fails on the '
                       'same graph with a negative dimension)'.format(self.subseq_len))

  def _encoder_strides(self):
    """advoc_model.py:90-116: (2,2) while the reference's running `n_time` (halved as a float per
    layer) is > 1, then (1,2): the time axis has collapsed, only frequency is down-sampled."""
    n_time = self.subseq_len / 2
    out = [(2, 2)]
    for _ in range(len(self._encoder_channels()) - 1):
      if n_time > 1:
        out.append((2, 2))
        n_time /= 2
      else:
        out.append((1, 2))
    return out

  def variable_specs(self):
    """TF variable names and shapes.  Generator entries are ordered decoder_1 .. decoder_N,
    encoder_N .. encoder_1 (the order their gradients complete in the backward pass)."""
    enc = self._encoder_channels()
    dec = self._decoder_specs()
    g = []
    prev = dec[-1][1] if dec else enc[-1]
    g.append(('generator/decoder_1/conv2d_transpose', (4, 4, 1, prev + enc[0])))
    for j in range(len(dec) - 1, -1, -1):
      idx, c, _ = dec[j]
      cin = enc[-1] if j == 0 else dec[j - 1][1] + enc[idx - 1]
      g.append(('generator/decoder_%d/conv2d_transpose' % idx, (4, 4, c, cin)))
    for i in range(len(enc) - 1, -1, -1):
      cin = 1 if i == 0 else enc[i - 1]
      g.append(('generator/encoder_%d/conv2d' % (i + 1), (4, 4, cin, enc[i])))
    d = []
    chans = [self.ndf, self.ndf * 2, self.ndf * 4, self.ndf * 8, 1]
    for i in range(4, -1, -1):
      cin = 2 if i == 0 else chans[i - 1]
      d.append(('discriminator/layer_%d/conv2d' % (i + 1), (4, 4, cin, chans[i])))

    def has_bn(scope):
      # batchnorm follows encoder_2.., every decoder but decoder_1, and layer_2..4
      # (advoc_model.py:118,142,194); never encoder_1 / decoder_1 / layer_1 / layer_5
      if not self.use_batchnorm:
        return False
      name = scope.split('/')[1]
      return name not in ('encoder_1', 'decoder_1', 'layer_1', 'layer_5')

    def expand(lst):
      out = []
      for scope, kshape in lst:
        cout = kshape[2] if 'transpose' in scope else kshape[3]
        out.append((scope + '/kernel', kshape))
        out.append((scope + '/bias', (cout,)))
        if has_bn(scope):
          base = scope.rsplit('/', 1)[0] + '/batch_normalization'
          out.append((base + '/gamma', (cout,)))
          out.append((base + '/beta', (cout,)))
      return out
    return expand(g), expand(d)

  # ------------------------------------------------------------------------------------------
  # build: allocate arenas + activation buffers, bind layers
  # ------------------------------------------------------------------------------------------
  def build(self, batch_size=None, seed=None, device=None):
    """Allocates parameters (N(0,0.02) kernels, zero biases: advoc_model.py:30,36) and the
    per-batch activation plan.  Called lazily by the first forward / train_loop."""
    self._check_supported()
    _lib.load()
    dev = device or _lib.device()
    if seed is not None:
      self._seed = seed
    B = int(batch_size or (self.train_batch_size if self.mode == Modes.TRAIN else self.eval_batch_size))
    if self._built is not None and self._built['B'] == B:
      return self
    st = self._built or {}
    gspec, dspec = self.variable_specs()
    if 'g_arena' not in st:
      st['g_arena'], st['d_arena'] = _Arena(gspec, dev), _Arena(dspec, dev)
      for net in ('g', 'd'):
        ar = st[net + '_arena']
        st[net + '_param'] = ar.new()
        st[net + '_grad'] = ar.new()
        st[net + '_m'] = ar.new()
        st[net + '_v'] = ar.new()
        st[net + '_P'] = ar.views(st[net + '_param'])
        st[net + '_G'] = ar.views(st[net + '_grad'])
      gen = torch.Generator().manual_seed(self._seed)
      for net in ('g', 'd'):
        for name, t in st[net + '_P'].items():
          if name.endswith('/kernel'):
            t.copy_(torch.randn(t.shape, generator=gen) * 0.02)
          elif name.endswith('/gamma'):
            t.fill_(1.0)
      st['g_t'] = st['d_t'] = 0
      st['sums'] = torch.zeros(4, dtype=torch.float32, device=dev)
      # largest |w| of every kernel, one launch per arena (advoc_segmented_amax_f32) at the start of each forward pass
      # instead of one magnitude pass per weight image (two per layer and step)
      for net in ('g', 'd'):
        ar = st[net + '_arena']
        names = [k for k in ar.offsets if k.endswith('/kernel')]
        sizes = [int(st[net + '_P'][k].numel()) for k in names]
        st[net + '_wamax_index'] = dict((k, i) for i, k in enumerate(names))
        st[net + '_wamax_off'] = torch.tensor([ar.offsets[k][0] for k in names], dtype=torch.int64, device=dev)
        st[net + '_wamax_size'] = torch.tensor(sizes, dtype=torch.int64, device=dev)
        st[net + '_wamax'] = torch.zeros(len(names), dtype=torch.int32, device=dev)
      st['wamax_on'] = os.environ.get('ADVOC_WEIGHT_AMAX', '1') == '1'
    st['B'] = B
    if 'side_stream' not in st:
      st['side_stream'] = torch.cuda.Stream(device=dev)
      st['side_on'] = os.environ.get('ADVOC_WGRAD_STREAM', '1') == '1'
    self._bind(st, B, dev)
    self._built = st
    if not hasattr(self, 'step'):
      self.step = 0
    self.G_vars = list(st['g_P'].keys())
    self.D_vars = list(st['d_P'].keys())
    return self

  def _bind(self, st, B, dev):
    f32 = dict(dtype=torch.float32, device=dev)
    T, F = self.subseq_len, self._nbins
    enc_c = self._encoder_channels()
    dec = self._decoder_specs()
    P, G = st['g_P'], st['g_G']
    bn_on = bool(self.use_batchnorm)
    st['bn_on'] = bn_on

    def new_bn(z, scope, PP, GG, scale=None, shift=None):
      """Per-tensor batch-norm state: the affine (scale, shift) is written by advoc_bn_forward and
      read by the CONSUMING layers' loads; scale/shift may alias a slice of a consumer's vector."""
      c = z.shape[3]
      base = scope + '/batch_normalization'
      return dict(z=z, c=c, npix=z.numel() // c, gamma=PP[base + '/gamma'], beta=PP[base + '/beta'],
                  dgamma=GG[base + '/gamma'], dbeta=GG[base + '/beta'],
                  scale=scale if scale is not None else torch.ones(c, **f32),
                  shift=shift if shift is not None else torch.zeros(c, **f32),
                  mean=torch.zeros(c, **f32), invstd=torch.ones(c, **f32),
                  work=torch.zeros(4 * c, **f32), copies=[])

    # ---- generator buffers ----
    # discriminator inputs for a 2B batch: [real ; fake]; G writes its output into the fake half.  The generator's input IS
    # the real half of the discriminator's conditioning input (the same tensor x in advoc_model.py:215-226): one copy less
    # per update
    st['d_cond'] = torch.zeros(2 * B, T, F, 1, **f32)
    st['x_in'] = st['d_cond'][:B]
    st['d_target'] = torch.zeros(2 * B, T, F, 1, **f32)
    gen_out = st['d_target'][B:]
    st['gen_out'] = gen_out
    e, ge = [], []
    h, w = T, F
    enc_s = self._encoder_strides()
    n_stride1 = sum(1 for st_ in enc_s if st_ == (1, 2))
    dec_s = [(1, 2) if j < n_stride1 else (2, 2) for j in range(len(dec))]     # advoc_model.py:139-142
    for c, (sh_, sw_) in zip(enc_c, enc_s):
      h, w = -(-h // sh_), -(-w // sw_)
      e.append(torch.zeros(B, h, w, c, **f32))
      ge.append(torch.zeros(B, h, w, c, **f32))
    st['enc'], st['g_enc'] = e, ge
    d, gd, masks = {}, {}, {}
    for j, (idx, c, drop) in enumerate(dec):
      src = e[-1] if j == 0 else d[dec[j - 1][0]]
      hh = src.shape[1] * dec_s[j][0]
      ww = (src.shape[2] if j == 0 else e[idx - 1].shape[2]) * 2
      d[idx] = torch.zeros(B, hh, ww, c, **f32)
      gd[idx] = torch.zeros(B, hh, ww, c, **f32)     # trimmed column is never written by backward
      if drop > 0:
        masks[idx] = (torch.zeros(B, hh, ww, c, dtype=torch.uint8, device=dev), 1.0 - drop)
    st['dec'], st['g_dec'], st['masks'] = d, gd, masks
    st['g_skip_amax'] = torch.zeros(len(e), dtype=torch.int32, device=dev)      # conv.Layer.backward_data(dx1_amax=...)

    # ---- generator batch-norm state (use_batchnorm=True) ----
    # encoder k >= 2 feeds encoder k+1 (alone) and decoder k (second half of a concat);
    # decoder idx feeds the next decoder (first half of the concat).
    gbn = collections.OrderedDict()
    dec_aff = {}   # decoder name -> (scale_vec, shift_vec) over its concatenated input channels
    if bn_on:
      for j, (idx, c, drop) in enumerate(dec):
        if j > 0:
          cin = d[dec[j - 1][0]].shape[3] + e[idx - 1].shape[3]
          dec_aff['decoder_%d' % idx] = (torch.ones(cin, **f32), torch.zeros(cin, **f32))
      if dec:
        cin = d[dec[-1][0]].shape[3] + e[0].shape[3]
        dec_aff['decoder_1'] = (torch.ones(cin, **f32), torch.zeros(cin, **f32))
      for i in range(1, len(enc_c)):
        gbn['encoder_%d' % (i + 1)] = new_bn(e[i], 'generator/encoder_%d' % (i + 1), P, G)
      for j, (idx, c, drop) in enumerate(dec):
        # this decoder's consumer: the next decoder in the list, or decoder_1 for the last one
        consumer = 'decoder_%d' % dec[j + 1][0] if j + 1 < len(dec) else 'decoder_1'
        sc, sh = dec_aff[consumer]
        gbn['decoder_%d' % idx] = new_bn(d[idx], 'generator/decoder_%d' % idx, P, G,
                                         scale=sc[:c], shift=sh[:c])
      # encoder outputs that are the second half of a decoder's concat: copy their affine over
      for j, (idx, c, drop) in enumerate(dec):
        if j > 0 and idx - 1 >= 1:
          c0 = d[dec[j - 1][0]].shape[3]
          sc, sh = dec_aff['decoder_%d' % idx]
          gbn['encoder_%d' % idx]['copies'].append((sc[c0:], sh[c0:]))
    st['g_bn'] = gbn

    def aff_of(name_src):
      b = gbn.get(name_src)
      return (b['scale'], b['shift']) if b else (None, None)

    def wamax(net, kernel_name):
      if not st['wamax_on']:
        return None
      i = st[net + '_wamax_index'][kernel_name]
      return st[net + '_wamax'][i:i + 1]

    # ---- generator layers ----
    L = collections.OrderedDict()
    x = st['x_in']
    for i, c in enumerate(enc_c):
      s = 'generator/encoder_%d/conv2d' % (i + 1)
      src = x if i == 0 else e[i - 1]
      pt, _ = C.same_pad(src.shape[1], 4, enc_s[i][0])
      pl, _ = C.same_pad(src.shape[2], 4, enc_s[i][1])
      sc, sh = aff_of('encoder_%d' % i) if i > 0 else (None, None)
      L['encoder_%d' % (i + 1)] = C.Layer(C.CONV, src, e[i], P[s + '/kernel'], P[s + '/bias'],
                                          w_amax=wamax('g', s + '/kernel'), stride=enc_s[i], pad=(pt, pl),
                                          in_act=C.ACT_NONE if i == 0 else C.ACT_LRELU,
                                          in_scale=sc, in_shift=sh)
    for j, (idx, c, drop) in enumerate(dec):
      s = 'generator/decoder_%d/conv2d_transpose' % idx
      if j == 0:
        x0, x1, in_w = e[-1], None, None
        sc, sh = aff_of('encoder_%d' % len(enc_c))
        src_mask = None
      else:
        x0, x1 = d[dec[j - 1][0]], e[idx - 1]
        in_w = x1.shape[2]                      # layers[-1][:, :, :-1, :]  (advoc_model.py:137)
        sc, sh = dec_aff.get('decoder_%d' % idx, (None, None))
        src_mask = masks.get(dec[j - 1][0])
      mk = masks.get(idx)
      # Without BN the producer applies its own dropout in its epilogue.  With BN dropout acts
      # AFTER the normalisation (advoc_model.py:142-149), i.e. on the consumer's loads.
      L['decoder_%d' % idx] = C.Layer(
          C.DECONV, x0, d[idx], P[s + '/kernel'], P[s + '/bias'], w_amax=wamax('g', s + '/kernel'), x1=x1, in_w=in_w,
          stride=dec_s[j],
          pad=(1, 1), in_act=C.ACT_RELU,
          drop_mask=mk[0] if (mk and not bn_on) else None, drop_scale=1.0 / mk[1] if (mk and not bn_on) else 0.,
          in_scale=sc, in_shift=sh,
          in_mask=src_mask[0] if (src_mask and bn_on) else None,
          in_mask_scale=1.0 / src_mask[1] if (src_mask and bn_on) else 0.)
    s = 'generator/decoder_1/conv2d_transpose'
    last = d[dec[-1][0]] if dec else e[-1]
    sc, sh = dec_aff.get('decoder_1', (None, None))
    src_mask = masks.get(dec[-1][0]) if dec else None
    L['decoder_1'] = C.Layer(C.DECONV, last, gen_out, P[s + '/kernel'], P[s + '/bias'],
                             w_amax=wamax('g', s + '/kernel'), x1=e[0],
                             in_w=e[0].shape[2], out_w=F, stride=(2, 2), pad=(1, 1), in_act=C.ACT_RELU,
                             in_scale=sc, in_shift=sh,
                             in_mask=src_mask[0] if (src_mask and bn_on) else None,
                             in_mask_scale=1.0 / src_mask[1] if (src_mask and bn_on) else 0.)
    st['g_layers'] = L
    if not bn_on:
      # who reads whose output: from the second step on a producer's forward epilogue writes its consumers' operand
      # images (conv.Layer.add_image_consumer; csrc/image_emit.h) -- encoder_i feeds encoder_{i+1} (leaky ReLU) and, as
      # the second concat source, a decoder (ReLU); a decoder feeds the next decoder's first source
      for i in range(1, len(enc_c)):
        L['encoder_%d' % i].add_image_consumer(L['encoder_%d' % (i + 1)], 0)
      for j, (idx, c, drop) in enumerate(dec):
        if j == 0:
          L['encoder_%d' % len(enc_c)].add_image_consumer(L['decoder_%d' % idx], 0)
        else:
          L['decoder_%d' % dec[j - 1][0]].add_image_consumer(L['decoder_%d' % idx], 0)
          L['encoder_%d' % idx].add_image_consumer(L['decoder_%d' % idx], 1)

    # ---- discriminator buffers + layers ----
    # BN off: the D step runs [real ; fake] as ONE 2B batch; BN on: two B passes (each pass has its
    # own batch statistics, as the reference's two build_discriminator calls do).
    chans = [self.ndf, self.ndf * 2, self.ndf * 4, self.ndf * 8, 1]
    strides = [2, 2, 2, 1, 1]
    a, ga = [], []
    h, w = T, F
    for c, s_ in zip(chans, strides):
      h, w = (h + 2 - 4) // s_ + 1, (w + 2 - 4) // s_ + 1
      a.append(torch.zeros(2 * B, h, w, c, **f32))
      ga.append(torch.zeros(2 * B, h, w, c, **f32))
    st['d_act'], st['g_d_act'] = a, ga
    DP, DG = st['d_P'], st['d_G']
    st['d_bn_scratch'] = (torch.zeros(max(chans), **f32), torch.zeros(max(chans), **f32))

    def d_layers(lo, hi):
      bns = {}
      if bn_on:
        for i in (1, 2, 3):
          bns[i] = new_bn(a[i][lo:hi], 'discriminator/layer_%d' % (i + 1), DP, DG)
      out = []
      for i in range(5):
        s = 'discriminator/layer_%d/conv2d' % (i + 1)
        if i == 0:
          lay = C.Layer(C.CONV, st['d_cond'][lo:hi], a[0][lo:hi], DP[s + '/kernel'], DP[s + '/bias'],
                        w_amax=wamax('d', s + '/kernel'), x1=st['d_target'][lo:hi], stride=(2, 2), pad=(1, 1), in_act=C.ACT_NONE)
        else:
          b = bns.get(i - 1)
          lay = C.Layer(C.CONV, a[i - 1][lo:hi], a[i][lo:hi], DP[s + '/kernel'], DP[s + '/bias'],
                        w_amax=wamax('d', s + '/kernel'), stride=(strides[i],) * 2, pad=(1, 1), in_act=C.ACT_LRELU,
                        in_scale=b['scale'] if b else None, in_shift=b['shift'] if b else None)
        out.append(lay)
      if not bn_on:
        for i in range(4):
          # (layer_1 .. layer_3: the next layer is the ONLY reader of the output -- where the kernels can, the output exists
          # as that layer's operand image only, conv.Layer.y_image_only; layer_4's reader, layer_5, reads fp32)
          out[i].add_image_consumer(out[i + 1], 0, exclusive=i < 3)
      return out, bns
    st['d_layers_fake'], st['d_bn_fake'] = d_layers(B, 2 * B)
    if bn_on:
      st['d_layers_real'], st['d_bn_real'] = d_layers(0, B)
    else:
      st['d_layers_2b'], _ = d_layers(0, 2 * B)
    st['g_d_target'] = torch.zeros(2 * B, T, F, 1, **f32)
    # inside the train step a layer's inputs do not change between its forward and its weight gradient, and backward_data /
    # backward_weight of a layer see the same gradient tensor: the operand images are made once per tensor and step
    for lay in list(L.values()) + st['d_layers_fake'] + st.get('d_layers_real', []) + st.get('d_layers_2b', []):
      lay.reuse_images = True
      lay.delayed_scale = True
    self._bind_weight_images(st, 'g', list(L.values()), dev)
    self._bind_weight_images(st, 'd', st['d_layers_fake'] + st.get('d_layers_real', []) + st.get('d_layers_2b', []), dev)

  def _bind_weight_images(self, st, net, layers, dev):
    """Persistent fp16 pair images of every kernel the image kernels read (forward and backward-data layouts), rebuilt
    by ONE advoc_weight_images_f32 launch per forward pass over the network (right after the magnitude launch) instead of
    one small launch per layer call -- 54 per train step.  Layers that share a kernel (the discriminator's real / fake /
    2B passes) share its images.  (Per-call images: ADVOC_WEIGHT_AMAX=0.)"""
    st[net + '_wimg'] = None
    st[net + '_wdirty'] = True            # new layers, new (empty) image pool
    if not st['wamax_on']:
      return
    names = dict((t.data_ptr(), k) for k, t in st[net + '_P'].items() if k.endswith('/kernel'))
    slots, rows, uses = {}, [], []
    pool_bytes = 0
    for lay in layers:
      name = names.get(lay.weight.data_ptr())
      if name is None:
        continue
      for direction in (0, 1):
        desc = lay.weight_image_desc(direction)
        if desc is None:
          continue
        key = (name,) + desc[:4]
        if key not in slots:
          slots[key] = (pool_bytes, len(rows))
          rows.append([st[net + '_arena'].offsets[name][0], desc[0], desc[1], desc[2], desc[3],
                       st[net + '_wamax_index'][name], pool_bytes, len(rows)])
          pool_bytes += desc[4]
        uses.append((lay, direction) + slots[key])
    if not rows:
      return
    pool = torch.empty(pool_bytes, dtype=torch.uint8, device=dev)
    # (r5) 32-word headers: {max |w|, 2^-s, taps, K, per-tap row-L1 maxima} (advoc_weight_images_l1_f32) -- the a-priori bounds of
    # the launches that write operand images from their epilogues read them
    hdrs = torch.zeros(32 * len(rows), dtype=torch.int32, device=dev)
    table = torch.tensor(rows, dtype=torch.int64, device=dev)
    for lay, direction, off, idx in uses:
      lay.set_weight_image(direction, pool.data_ptr() + off, hdrs.data_ptr() + 128 * idx, l1=True)
    st[net + '_wimg'] = dict(pool=pool, hdrs=hdrs, table=table, count=len(rows), uses=uses)

  # ------------------------------------------------------------------------------------------
  # batch-norm plumbing
  # ------------------------------------------------------------------------------------------
  def _bn_forward(self, b):
    lib = _lib.load()
    if self._sync_bn is None:
      _lib.check(lib.advoc_bn_forward(
          _lib.ptr(b['z']), b['npix'], b['c'], _lib.ptr(b['gamma']), _lib.ptr(b['beta']), 1e-5,
          _lib.ptr(b['scale']), _lib.ptr(b['shift']), _lib.ptr(b['mean']), _lib.ptr(b['invstd']),
          _lib.ptr(b['work']), _lib.stream()), 'advoc_bn_forward')
    else:
      # statistics over the GLOBAL batch: (sum z, sum z^2) in double, summed across the replicas
      _lib.check(lib.advoc_bn_forward_stats(_lib.ptr(b['z']), b['npix'], b['c'], _lib.ptr(b['work']), _lib.stream()),
                 'advoc_bn_forward_stats')
      self._sync_bn(b['work'].view(torch.float64)[:2 * b['c']])
      _lib.check(lib.advoc_bn_forward_finalize(
          _lib.ptr(b['work']), b['npix'] * self._world_size, b['c'], _lib.ptr(b['gamma']), _lib.ptr(b['beta']), 1e-5,
          _lib.ptr(b['scale']), _lib.ptr(b['shift']), _lib.ptr(b['mean']), _lib.ptr(b['invstd']), _lib.stream()),
          'advoc_bn_forward_finalize')
    for sc, sh in b['copies']:
      sc.copy_(b['scale'])
      sh.copy_(b['shift'])

  def _bn_backward(self, b, g, accumulate=False, discard_param_grads=False):
    st = self._built
    lib = _lib.load()
    dg, db = (st['d_bn_scratch'][0][:b['c']], st['d_bn_scratch'][1][:b['c']]) if discard_param_grads \
        else (b['dgamma'], b['dbeta'])
    if self._sync_bn is None:
      _lib.check(lib.advoc_bn_backward(
          _lib.ptr(b['z']), _lib.ptr(g), b['npix'], b['c'], _lib.ptr(b['gamma']), _lib.ptr(b['mean']),
          _lib.ptr(b['invstd']), _lib.ptr(dg), _lib.ptr(db), int(accumulate), _lib.ptr(b['work']),
          _lib.stream()), 'advoc_bn_backward')
      return
    _lib.check(lib.advoc_bn_backward_stats(
        _lib.ptr(b['z']), _lib.ptr(g), b['npix'], b['c'], _lib.ptr(b['mean']), _lib.ptr(b['invstd']),
        _lib.ptr(dg), _lib.ptr(db), int(accumulate), _lib.ptr(b['work']), _lib.stream()), 'advoc_bn_backward_stats')
    self._sync_bn(b['work'].view(torch.float64)[:2 * b['c']])
    _lib.check(lib.advoc_bn_backward_apply(
        _lib.ptr(b['z']), _lib.ptr(g), b['npix'], b['c'], _lib.ptr(b['gamma']), _lib.ptr(b['mean']),
        _lib.ptr(b['invstd']), _lib.ptr(b['work']), b['npix'] * self._world_size, _lib.stream()),
        'advoc_bn_backward_apply')

  # ------------------------------------------------------------------------------------------
  # parameters in / out (TF variable names)
  # ------------------------------------------------------------------------------------------
  def state_dict(self):
    self.build()
    self._flush_d_adam()
    out = collections.OrderedDict()
    for net in ('g', 'd'):
      for k, v in self._built[net + '_P'].items():
        out[k] = v.detach().clone()
    out['global_step'] = torch.tensor(self.step)
    return out

  def load_state_dict(self, state):
    self.build()
    self._flush_d_adam()
    for net in ('g', 'd'):
      for k, v in self._built[net + '_P'].items():
        if k in state:
          v.copy_(state[k].to(v.device, torch.float32))
    self.parameters_changed()
    if 'global_step' in state:
      self.step = int(state['global_step'])

  def parameters_changed(self, net=None):
    """The kernels of `net` ('g' / 'd', None = both) were written: their magnitude table and weight images are rebuilt by
    the next forward pass over the network.  The framework's own writers (Adam, load_state_dict, the data-parallel
    broadcast) call this; code that edits the parameter views in place must call it too."""
    if self._built:
      for n in (('g', 'd') if net is None else (net,)):
        self._built[n + '_wdirty'] = True

  def _image_header_sum(self, word):
    st = self._built
    if not st:
      return 0
    layers = list(st.get('g_layers', {}).values()) + st.get('d_layers_fake', []) + st.get('d_layers_real', []) \
        + st.get('d_layers_2b', [])
    hdrs = [h for lay in layers for h in lay.image_headers()]
    if not hdrs:
      return 0
    return int(torch.stack([h[word].to(torch.int64) for h in hdrs]).sum().item())

  def image_saturations(self):
    """Elements that left the fp16 head room of a one-pass (delayed-scale) operand image, summed over all layers since
    the buffers were allocated: a tensor grew more than 64 x from one step to the next.  Every such image was rebuilt
    with its exact scale before anything read it (image_refits), so this is a statistic, not an error.  Synchronises;
    call it at summary time, not per step."""
    return self._image_header_sum(3)

  def image_refits(self):
    """Operand images that the device re-built with the exact scale because the tensor left the [2^-6, 2^6] window of
    its previous magnitude (csrc/image.hip: refit_image_kernel).  Costs one extra pass over that tensor each."""
    return self._image_header_sum(5)

  def optimizer_state(self):
    self._flush_d_adam()
    st = self._built
    return {k: st[k].detach().clone() for k in ('g_m', 'g_v', 'd_m', 'd_v')}, (st['g_t'], st['d_t'])

  def load_optimizer_state(self, tensors, steps):
    self._flush_d_adam()
    st = self._built
    for k in ('g_m', 'g_v', 'd_m', 'd_v'):
      st[k].copy_(tensors[k])
    st['g_t'], st['d_t'] = steps

  # ------------------------------------------------------------------------------------------
  # dropout
  # ------------------------------------------------------------------------------------------
  def set_dropout_masks(self, masks):
    """Inject {0,1} masks {'decoder_N': tensor like the layer output} (parity tests); None
    returns to the on-device Philox stream."""
    self._injected_masks = masks

  def _refresh_masks(self, clip_offset=0):
    st = self._built
    if self._injected_masks is not None:
      for idx, (buf, keep) in st['masks'].items():
        buf.copy_(self._injected_masks['decoder_%d' % idx].to(buf.device).to(torch.uint8))
      return
    lib = _lib.load()
    self._dropout_calls += 1
    for idx, (buf, keep) in st['masks'].items():
      per_clip = buf[0].numel()
      seed = (self._seed * 1000003 + idx) * 2654435761 + self._dropout_calls
      _lib.check(lib.advoc_dropout_mask_u8(_lib.ptr(buf), buf.numel(), seed & (2 ** 64 - 1),
                                           (clip_offset * per_clip + 3) // 4 * 4, keep, _lib.stream()),
                 'advoc_dropout_mask_u8')

  # ------------------------------------------------------------------------------------------
  # forward passes
  # ------------------------------------------------------------------------------------------
  def _refresh_weight_amax(self, net):
    """max |w| of every kernel of arena `net` ('g' / 'd') and every weight image of the network, one launch each.  Called
    at the start of each forward pass over the network; does its work when the parameters were written since the last
    time (parameters_changed: Adam step, load_state_dict, broadcast) -- i.e. ONCE per optimizer step; the second generator
    pass of a train_loop and the backward passes read the same images."""
    st = self._built
    if not st['wamax_on'] or not st.get(net + '_wdirty', True):
      return
    st[net + '_wdirty'] = False
    out = st[net + '_wamax']
    _lib.check(_lib.load().advoc_segmented_amax_f32(
        _lib.ptr(st[net + '_param']), _lib.ptr(st[net + '_wamax_off']), _lib.ptr(st[net + '_wamax_size']),
        out.numel(), _lib.ptr(out), _lib.stream()), 'advoc_segmented_amax_f32')
    wi = st.get(net + '_wimg')
    if wi:
      _lib.check(_lib.load().advoc_weight_images_l1_f32(
          _lib.ptr(st[net + '_param']), _lib.ptr(out), _lib.ptr(wi['table']), wi['count'], _lib.ptr(wi['pool']),
          _lib.ptr(wi['hdrs']), _lib.stream()), 'advoc_weight_images_l1_f32')

  def _gen_forward(self, x):
    st = self._built
    if x.data_ptr() != st['x_in'].data_ptr():
      st['x_in'].copy_(x)
    self._refresh_weight_amax('g')
    self._refresh_masks(self._rank * st['B'])
    gbn = st['g_bn']
    for name, lay in st['g_layers'].items():
      lay.forward()
      if name in gbn:
        self._bn_forward(gbn[name])
    return st['gen_out']

  def _disc_forward(self, layers, bns):
    self._refresh_weight_amax('d')
    for i, lay in enumerate(layers):
      lay.forward()
      if i in bns:
        self._bn_forward(bns[i])

  def build_generator(self, x):
    """x: [B, subseq_len, 513, 1] float32 -> generated magnitude spectrogram, same shape
    (advoc_model.py:75-166).  Dropout is active in every mode, as in the reference."""
    x = x if isinstance(x, torch.Tensor) else torch.as_tensor(x)
    x = x.to(_lib.device(), torch.float32)
    self.build(batch_size=x.shape[0])
    return self._gen_forward(x).clone()

  def build_discriminator(self, discrim_inputs, discrim_targets):
    """[B,T,513,1] x2 -> patch probabilities [B, 30, 62, 1] (advoc_model.py:168-204)."""
    dev = _lib.device()
    cond = torch.as_tensor(discrim_inputs).to(dev, torch.float32)
    tgt = torch.as_tensor(discrim_targets).to(dev, torch.float32)
    self.build(batch_size=cond.shape[0])
    self._flush_d_adam()
    st = self._built
    B = st['B']
    st['d_cond'][B:].copy_(cond)
    st['d_target'][B:].copy_(tgt)
    self._disc_forward(st['d_layers_fake'], st['d_bn_fake'])
    logits = st['d_act'][4][B:]
    prob = torch.empty_like(logits)
    _lib.check(_lib.load().advoc_sigmoid_f32(_lib.ptr(logits), _lib.ptr(prob), logits.numel(), _lib.stream()),
               'advoc_sigmoid_f32')
    return prob

  # ------------------------------------------------------------------------------------------
  # train step
  # ------------------------------------------------------------------------------------------
  def __call__(self, x, target=None, x_wav=None, x_mel_spec=None):
    """Wires the model to its inputs (the reference builds the loss / optimiser graph here,
    advoc_model.py:206-281).  `x` is either a callable returning a fresh
    (x_inverted, x_magspec, x_wav, x_melspec) tuple per call -- the analogue of the reference's
    iterator-backed tensors, each train op pulling a NEW batch -- or a fixed batch
    (x, target[, x_wav, x_mel_spec]) of device tensors that every step reuses."""
    if callable(x):
      self._feed = x
    else:
      fixed = tuple(x) if isinstance(x, (tuple, list)) else (x, target, x_wav, x_mel_spec)
      self._feed = lambda: fixed
    return self

  def _adam(self, net, reduced=False):
    st = self._built
    st[net + '_t'] += 1
    t = st[net + '_t']
    lr_t = self._lr * math.sqrt(1 - self._beta2 ** t) / (1 - self._beta1 ** t)
    flat = st[net + '_grad']
    if reduced:
      pass                                               # (_flush_d_adam: the cross-rank sum was started earlier and joined)
    elif net == 'g' and self._reduce_async is not None:
      start, finish, _ = self._reduce_async
      start(flat, st.get('g_sent', 0), flat.numel())     # whatever the backward pass has not sent yet
      finish()
      st['g_sent'] = 0
    elif self._allreduce is not None:
      self._allreduce(flat)
    _lib.check(_lib.load().advoc_adam_tf_f32(
        _lib.ptr(st[net + '_param']), _lib.ptr(flat), _lib.ptr(st[net + '_m']), _lib.ptr(st[net + '_v']),
        flat.numel(), lr_t, self._beta1, self._beta2, self._adam_eps, 1.0 / self._world_size,
        _lib.stream()), 'advoc_adam_tf_f32')
    st[net + '_wdirty'] = True

  def _zero_arena(self, net):
    flat = self._built[net + '_grad']
    _lib.check(_lib.load().advoc_zero_f32(_lib.ptr(flat), flat.numel(), _lib.stream()), 'advoc_zero_f32')

  def _load_batch(self, batch):
    st = self._built
    B = st['B']
    x, target = batch[0], batch[1]
    st['x_in'].copy_(x)                # (= d_cond[:B])
    st['d_cond'][B:].copy_(x)
    st['d_target'][:B].copy_(target)
    st['last_batch'] = batch

  def media_summaries(self, max_outputs=3):
    """The image / audio summaries of advoc_model.py:258-281 for the batch of the last step: (images, audio), both
    {name: numpy array}.  Images are [n, 513 (high frequencies on top), T] -- tf.image.rot90 of [n, T, F, 1]; audio is the
    phase reconstruction (SpectralUtil.audio_from_mag_spec) of clip 0's input / target / generated spectrogram plus the
    source waveform.  Synchronises and runs the vocoder back end: call it at summary time only."""
    import numpy as np
    st = self._built
    if not st or 'last_batch' not in st:
      return {}, {}
    B = st['B']
    n = min(max_outputs, B)

    def rot90(t):                      # [n, T, F, 1] -> [n, F, T], counter-clockwise: row 0 = last column
      return np.ascontiguousarray(np.flip(t[:n, :, :, 0].detach().cpu().numpy().transpose(0, 2, 1), axis=1))
    x, target, gen = st['x_in'], st['d_target'][:B], st['gen_out']
    images = dict(input_magspec=rot90(x), generated_magspec=rot90(gen), target_magspec=rot90(target))
    batch = st['last_batch']
    if len(batch) > 3 and batch[3] is not None:
      images['input_melspec'] = rot90(torch.as_tensor(batch[3]))
    audio = {}
    if getattr(self, 'spectral', None) is None:
      from advoc_amd.spectral_util import SpectralUtil
      self.spectral = SpectralUtil()
    for name, t in (('input_audio', x), ('target_audio', target), ('gen_audio', gen)):
      audio[name] = np.asarray(self.spectral.audio_from_mag_spec(t[0].detach().cpu().numpy()), dtype=np.float32).reshape(1, -1)
    if len(batch) > 2 and batch[2] is not None:
      audio['target_x_wav'] = torch.as_tensor(batch[2])[:n].detach().cpu().numpy().reshape(n, -1)
    return images, audio

  def d_step(self, batch):
    """One discriminator update on `batch` (advoc_model.py:238,257): G forward, D on real and on
    fake (one 2B batch without BN, two B passes with BN), discrim_loss, D weight gradients, Adam."""
    st = self._built
    lib = _lib.load()
    B = st['B']
    self._flush_d_adam()
    self._load_batch(batch)
    self._gen_forward(st['x_in'])
    if st['bn_on']:
      passes = [(st['d_layers_real'], st['d_bn_real'], 0, B), (st['d_layers_fake'], st['d_bn_fake'], B, 2 * B)]
    else:
      passes = [(st['d_layers_2b'], {}, 0, 2 * B)]
    for layers, bns, lo, hi in passes:
      self._disc_forward(layers, bns)
    logits = st['d_act'][4]
    glog = st['g_d_act'][4]
    n = logits[:B].numel()
    _lib.check(lib.advoc_gan_d_loss(_lib.ptr(logits[:B]), _lib.ptr(logits[B:]), n, _lib.ptr(glog[:B]),
                                    _lib.ptr(glog[B:]), _lib.ptr(st['sums'][0:1]), _lib.stream()),
               'advoc_gan_d_loss')
    DG = st['d_G']
    # one fill of the whole gradient arena instead of one small memset per kernel / bias / BN vector
    # (every weight-gradient kernel accumulates with atomics into zeroed memory anyway)
    self._zero_arena('d')
    for k, (layers, bns, lo, hi) in enumerate(passes):
      acc = True
      for lay in layers:
        lay.set_dy_role('d')
      for i in range(4, -1, -1):
        s = 'discriminator/layer_%d/conv2d' % (i + 1)
        g = st['g_d_act'][i][lo:hi]
        if i in bns:
          self._bn_backward(bns[i], g, accumulate=acc)
        # backward-data first: it leaves the fp16 pair image of g behind, which the weight gradient reads again
        if i > 0:
          # (layer_5 -> layer_4 without batch norm in between: layer_5's epilogue writes layer_4's output-gradient image and
          # its bias gradient -- the largest of the image passes, conv.Layer.backward_data)
          # (r5: every link of the chain -- conv.Layer._dx_target takes the ones whose kernels can: layer_4 -> layer_3 and
          # layer_3 -> layer_2 on the patch kernels under the a-priori scale, image only)
          below = layers[i - 1] if (i - 1) not in bns else None
          layers[i].backward_data(g, st['g_d_act'][i - 1][lo:hi], db=DG[s + '/bias'], db_accumulate=acc,
                                  grad_consumer=below,
                                  consumer_db=DG['discriminator/layer_%d/conv2d/bias' % i] if below is not None else None,
                                  consumer_db_accumulate=acc)
        with self._wgrad_ctx():
          layers[i].backward_weight(g, DG[s + '/kernel'], DG[s + '/bias'], accumulate=acc)
      # (batch norm, two passes: r3 joined the side stream here because pass 1's backward-data launches gave wrong elements
      # next to pass 0's thin weight gradient.  The cause was a missing wait in the LDS-DMA kernels -- their fragment reads
      # could still be queued when the next tile's DMA overwrote the stage, exposed by that kernel's LDS atomics; fixed in
      # csrc/lds_dma.h (dma_ring_barrier), so the passes overlap again: tests/test_hip_model.py, side stream vs serial.)
    self._join_wgrad()
    if self._reduce_async is not None:
      # data parallel: the discriminator's gradient sum (11 MB) starts now and runs, on RCCL's stream, under the generator
      # forward of the G step that follows; its Adam step is applied when something needs the discriminator's parameters
      # (_flush_d_adam: before D(fake) in g_step, before the next d_step, before any read of the parameters)
      self._reduce_async[0](st['d_grad'], 0, st['d_grad'].numel())
      st['d_adam_pending'] = True
    else:
      self._adam('d')
    st['last_counts_d'] = n

  def _flush_d_adam(self):
    st = self._built
    if st and st.pop('d_adam_pending', False):
      self._reduce_async[1]()
      self._adam('d', reduced=True)

  def _wgrad_ctx(self):
    """Weight / bias gradients are off the critical path of the backward pass (nothing downstream
    reads them before Adam): they run on a side stream so that their launches fill the issue slots
    and tails the backward-data kernels leave idle, and vice versa (measured -3.3 % step time;
    ADVOC_WGRAD_STREAM=0 keeps everything on one stream).  Concurrent kernels share the CUs, which
    makes per-kernel durations uninterpretable: while a launch profiler is attached
    (conv.Layer.profiler, bench.py's instrumented steps) the step runs serially.  Returns a context
    manager."""
    st = self._built
    if not st.get('side_on', False) or C.Layer.profiler is not None:
      return contextlib.nullcontext()
    side = st['side_stream']
    side.wait_stream(torch.cuda.current_stream())       # everything enqueued so far (dy is ready)
    return torch.cuda.stream(side)

  def _join_wgrad(self):
    st = self._built
    if st.get('side_on', False):
      torch.cuda.current_stream().wait_stream(st['side_stream'])

  def _last_param_of(self, scope):
    """Name of the arena entry that ends `scope`'s block of parameters."""
    names = [k for k in self._built['g_arena'].offsets if k.startswith(scope + '/')]
    return names[-1]

  def _g_grads_ready(self, last_name):
    """Called when the generator backward has finished every parameter up to and including
    `last_name` (arena order = backward-completion order): starts the cross-rank sum of the newly
    completed arena range once it is at least one bucket long."""
    if self._reduce_async is None:
      return
    st = self._built
    start, _, bucket = self._reduce_async
    off, shape = st['g_arena'].offsets[last_name]
    n = 1
    for d_ in shape:
      n *= d_
    hi = (off + n + 3) // 4 * 4
    lo = st.get('g_sent', 0)
    if hi - lo >= bucket:
      start(st['g_grad'], lo, hi)
      st['g_sent'] = hi

  def g_step(self, batch):
    """One generator update on `batch` (advoc_model.py:239-245,254-255): G forward, D(fake)
    forward, gen_loss, backward through D to the generator output, G backward, Adam."""
    st = self._built
    lib = _lib.load()
    B = st['B']
    self._load_batch(batch)
    gen = self._gen_forward(st['x_in'])
    self._flush_d_adam()                    # (data parallel: the D update of the d_step before, its gradient sum joined here)
    use_gan = self.gan_weight > 0
    g_out = st['g_d_target'][B:]            # gradient w.r.t. the generator output
    logits = st['d_act'][4][B:]
    glog = st['g_d_act'][4][B:]
    Lf, bnf = st['d_layers_fake'], st['d_bn_fake']
    if use_gan:
      self._disc_forward(Lf, bnf)
    _lib.check(lib.advoc_gan_g_loss(
        _lib.ptr(logits) if use_gan else None, logits.numel(), _lib.ptr(gen), _lib.ptr(st['d_target'][:B]),
        gen.numel(), float(self.gan_weight), float(self.l1_weight), _lib.ptr(glog) if use_gan else None,
        _lib.ptr(g_out), 0, _lib.ptr(st['sums'][1:3]), _lib.stream()), 'advoc_gan_g_loss')
    if use_gan:
      for lay in Lf:
        lay.set_dy_role('g')      # G-loss gradients: their own magnitude history (the D step's are ~sigma(D(fake)) x smaller)
      for i in range(4, 0, -1):
        if i in bnf:   # through the discriminator's batch norm; its parameter gradients are not used here
          self._bn_backward(bnf[i], st['g_d_act'][i][B:], discard_param_grads=True)
        Lf[i].backward_data(st['g_d_act'][i][B:], st['g_d_act'][i - 1][B:],
                            grad_consumer=Lf[i - 1] if (i - 1) not in bnf else None)
      Lf[0].backward_data(st['g_d_act'][0][B:], None, g_out, accum1=True)
    # generator backward: decoder_1 .. decoder_N, then encoder_N .. encoder_1
    GL, GG = st['g_layers'], st['g_G']
    dec = self._decoder_specs()
    e, ge, gd = st['enc'], st['g_enc'], st['g_dec']
    gbn = st['g_bn']
    if st['bn_on']:
      # BN backward rewrites the whole gradient tensor in place, including the trimmed column the
      # consumers never write: start every step from zero there
      for t in gd.values():
        t.zero_()
    s = 'generator/decoder_1/conv2d_transpose'
    st['g_sent'] = 0
    self._zero_arena('g')      # one fill for the whole arena; the kernels below accumulate into it
    last_idx = dec[-1][0] if dec else None
    # (r5) every decoder hands the layer that produced its first source that layer's output-gradient IMAGE (and bias sums)
    # instead of the fp32 tensor where its backward-data kernel can (conv.Layer._dx_target: no batch norm, no dropout mask
    # in between)
    def below(idx_):
      name_ = 'decoder_%d' % idx_
      if name_ in gbn:
        return None, None
      return GL[name_], GG['generator/%s/conv2d_transpose/bias' % name_]
    skip_amax = st['g_skip_amax']
    skip_amax.zero_()
    skip_tracked = set()
    lower, lower_db = below(last_idx) if dec else (None, None)
    GL['decoder_1'].backward_data(g_out, gd[last_idx] if dec else ge[-1], ge[0], grad_consumer=lower, consumer_db=lower_db)
    with self._wgrad_ctx():
      GL['decoder_1'].backward_weight(g_out, GG[s + '/kernel'], GG[s + '/bias'], accumulate=True)
      self._g_grads_ready(self._last_param_of('generator/decoder_1'))
    for j in range(len(dec) - 1, -1, -1):
      idx = dec[j][0]
      s = 'generator/decoder_%d/conv2d_transpose' % idx
      lay = GL['decoder_%d' % idx]
      if 'decoder_%d' % idx in gbn:
        self._bn_backward(gbn['decoder_%d' % idx], gd[idx], accumulate=True)
      if j == 0:
        lay.backward_data(gd[idx], ge[-1], db=GG[s + '/bias'])
      else:
        lower, lower_db = below(dec[j - 1][0])
        # (skip_amax[idx - 1]: the largest |skip gradient| written to ge[idx - 1] -- what the encoder's accumulating
        # backward-data call below needs to bound the sum whose image it writes)
        lay.backward_data(gd[idx], gd[dec[j - 1][0]], ge[idx - 1], db=GG[s + '/bias'], grad_consumer=lower,
                          consumer_db=lower_db, dx1_amax=skip_amax[idx - 1:idx])
        if lay.tracks_dx1_amax():
          skip_tracked.add(idx - 1)
      with self._wgrad_ctx():
        lay.backward_weight(gd[idx], GG[s + '/kernel'], GG[s + '/bias'], accumulate=True)
        self._g_grads_ready(self._last_param_of('generator/decoder_%d' % idx))
    for i in range(len(e) - 1, -1, -1):
      s = 'generator/encoder_%d/conv2d' % (i + 1)
      lay = GL['encoder_%d' % (i + 1)]
      if 'encoder_%d' % (i + 1) in gbn:
        self._bn_backward(gbn['encoder_%d' % (i + 1)], ge[i], accumulate=True)
      if i > 0:
        # (r5) ge[i - 1] holds the decoder's skip gradient; this call adds the encoder path's and -- where the kernels can and the
        # skip gradient's magnitude was recorded -- leaves the SUM as encoder_i's output-gradient image only
        name_b = 'encoder_%d' % i
        lower = GL[name_b] if (name_b not in gbn and (i - 1) in skip_tracked) else None
        lay.backward_data(ge[i], ge[i - 1], accum0=True, db=GG[s + '/bias'], grad_consumer=lower,
                          consumer_db=GG['generator/%s/conv2d/bias' % name_b] if lower is not None else None,
                          bound_add=skip_amax[i - 1:i] if lower is not None else None)
      with self._wgrad_ctx():
        lay.backward_weight(ge[i], GG[s + '/kernel'], GG[s + '/bias'], accumulate=True)
        self._g_grads_ready(self._last_param_of('generator/encoder_%d' % (i + 1)))
    self._join_wgrad()
    self._adam('g')
    self.step += 1
    st['last_counts_g'] = (logits.numel(), gen.numel())

  def G_train_op(self, batch=None):
    """The reference's `G_train_op` (advoc_model.py:254-255: Adam minimize of gen_loss_total over G_vars, increments the
    global step) as a callable: one generator update on `batch`, or on the next batch of the feed given to __call__
    (what `sess.run(model.G_train_op)` does with the iterator-backed inputs)."""
    if batch is None:
      if self._feed is None:
        raise RuntimeError('call model(feed) first')
      batch = self._feed()
    self.build(batch_size=batch[0].shape[0])
    self.g_step(batch)
    return self.step

  def D_train_op(self, batch=None):
    """The reference's `D_train_op` (advoc_model.py:256-257): one discriminator update, see G_train_op."""
    if batch is None:
      if self._feed is None:
        raise RuntimeError('call model(feed) first')
      batch = self._feed()
    self.build(batch_size=batch[0].shape[0])
    self.d_step(batch)
    # (data parallel: inside train_loop the D update is applied after the generator forward of the G step, under which its
    # gradient sum runs; a caller of the op on its own gets the updated parameters back, whatever it reads next -- ADVICE r4)
    self._flush_d_adam()

  def train_loop(self, sess=None):
    """D update on one batch, G update on the NEXT batch; returns the global step
    (reference advoc_model.py:285-289; `sess` is accepted and ignored)."""
    if self._feed is None:
      raise RuntimeError('call model(feed) first')
    batch = self._feed()
    self.build(batch_size=batch[0].shape[0])
    if self.gan_weight > 0:
      self.d_step(batch)
      batch = self._feed()
    self.g_step(batch)
    return self.step

  def losses(self):
    """Last step's loss scalars (one device->host sync): the tf.summary.scalar values of
    advoc_model.py:272-275."""
    st = self._built
    s = st['sums'].cpu()
    out = {}
    if 'last_counts_d' in st:
      out['disc_loss'] = float(s[0]) / st['last_counts_d']
    if 'last_counts_g' in st:
      nl, ns = st['last_counts_g']
      out['gen_loss_GAN'] = float(s[1]) / nl
      out['gen_loss_L1'] = float(s[2]) / ns
      out['gen_loss_total'] = (out['gen_loss_GAN'] * self.gan_weight if self.gan_weight > 0 else 0.) \
          + out['gen_loss_L1'] * self.l1_weight
    return out

  def l1_eval(self, x, target):
    """mean |target - G(x)| (train_evaluate.py:137, eval mode)."""
    gen = self.build_generator(x)
    st = self._built
    tgt = torch.as_tensor(target).to(gen.device, torch.float32).contiguous()
    _lib.check(_lib.load().advoc_gan_g_loss(None, 0, _lib.ptr(gen), _lib.ptr(tgt), gen.numel(), 0.0, 1.0,
                                            None, None, 0, _lib.ptr(st['sums'][1:3]), _lib.stream()),
               'advoc_gan_g_loss')
    return float(st['sums'][2].cpu()) / gen.numel(), gen


class AdvocSmall(Advoc):
  """advoc_model_small.py: ngf = ndf = 32, encoder_1 + 4 encoders, dropout on decoder_5/4."""
  ngf = 32
  ndf = 32
  num_enc_layers = 4
  _dec_spec_small = [(8, 0.5), (8, 0.5), (8, 0.5), (8, 0.5), (4, 0.5), (2, 0.0), (1, 0.0)]

  def _encoder_channels(self):
    return [self.ngf] + [self.ngf * m for m in self._enc_mult[:self.num_enc_layers]]

  def _decoder_specs(self):
    spec = self._dec_spec_small[len(self._dec_spec_small) - self.num_enc_layers:]
    n_enc = 1 + self.num_enc_layers
    return [(n_enc - i, self.ngf * m, d) for i, (m, d) in enumerate(spec)]
