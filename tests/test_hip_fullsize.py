"""Size-independent properties at BASELINE.json's full sizes (configs[1]: AdVoc-small, 32 clips of 256 x 513 per
GPU, the discriminator's 2B = 64 batch; configs[2]: AdVoc-full, 64 clips, 2B = 128), where the CPU oracle would take
minutes per layer:

  * adjointness  <dy, conv(x)> == <conv_backward_data(dy), x> == <conv_backward_weight(dy), w>
    for every distinct layer shape of the two networks (bias 0, identity activation: the three
    directions are then three views of ONE trilinear form, so each inner product must agree to
    fp32 round-off whatever the kernels' tiling, split-K or two-stage paths do),
  * positive homogeneity of the fused leaky-ReLU prologue, conv(2x) - b == 2 (conv(x) - b),
  * iSTFT(STFT(x)) == x away from the clip ends for a full training batch of waveforms,
  * the Philox dropout stream does not depend on how a batch is sharded.
The small-shape oracle comparisons live in test_hip_conv.py / test_hip_model.py."""
import collections

import pytest
import torch

gpu = pytest.mark.gpu
B = 32

# name, kind, (batch, H, W), c0, c1, cout, trim, stride, pad
LAYERS = [
    ('encoder_1', 0, (B, 256, 513), 1, 0, 32, 0, (2, 2), (1, 1)),
    ('encoder_2', 0, (B, 128, 257), 32, 0, 64, 0, (2, 2), (1, 1)),
    ('encoder_3', 0, (B, 64, 129), 64, 0, 128, 0, (2, 2), (1, 1)),
    ('encoder_4', 0, (B, 32, 65), 128, 0, 256, 0, (2, 2), (1, 1)),
    ('encoder_5', 0, (B, 16, 33), 256, 0, 256, 0, (2, 2), (1, 1)),
    ('decoder_5', 1, (B, 8, 17), 256, 0, 256, 0, (2, 2), (1, 1)),
    ('decoder_4', 1, (B, 16, 33), 256, 256, 128, 1, (2, 2), (1, 1)),
    ('decoder_3', 1, (B, 32, 65), 128, 128, 64, 1, (2, 2), (1, 1)),
    ('decoder_2', 1, (B, 64, 129), 64, 64, 32, 1, (2, 2), (1, 1)),
    ('decoder_1', 1, (B, 128, 257), 32, 32, 1, 1, (2, 2), (1, 1)),
    ('layer_1', 0, (2 * B, 256, 513), 1, 1, 32, 0, (2, 2), (1, 1)),
    ('layer_2', 0, (2 * B, 128, 256), 32, 0, 64, 0, (2, 2), (1, 1)),
    ('layer_3', 0, (2 * B, 64, 128), 64, 0, 128, 0, (2, 2), (1, 1)),
    ('layer_4', 0, (2 * B, 32, 64), 128, 0, 256, 0, (1, 1), (1, 1)),
    ('layer_5', 0, (2 * B, 31, 63), 256, 0, 1, 0, (1, 1), (1, 1)),
]


# BASELINE configs[2]: AdVoc (full, ngf = ndf = 64, 8 encoders) at 64 clips per GPU -- the shapes of
# profiles/r01_n_layer_times.md §full; the kernel-selection rules pick different instances here than at B = 1
BF = 64
LAYERS_FULL = [
    ('encoder_1', 0, (BF, 256, 513), 1, 0, 64, 0, (2, 2), (1, 1)),
    ('encoder_2', 0, (BF, 128, 257), 64, 0, 128, 0, (2, 2), (1, 1)),
    ('encoder_3', 0, (BF, 64, 129), 128, 0, 256, 0, (2, 2), (1, 1)),
    ('encoder_4', 0, (BF, 32, 65), 256, 0, 512, 0, (2, 2), (1, 1)),
    ('encoder_5', 0, (BF, 16, 33), 512, 0, 512, 0, (2, 2), (1, 1)),
    ('encoder_6', 0, (BF, 8, 17), 512, 0, 512, 0, (2, 2), (1, 1)),
    ('encoder_7', 0, (BF, 4, 9), 512, 0, 512, 0, (2, 2), (1, 1)),
    ('encoder_8', 0, (BF, 2, 5), 512, 0, 512, 0, (2, 2), (1, 1)),
    ('decoder_8', 1, (BF, 1, 3), 512, 0, 512, 0, (2, 2), (1, 1)),
    ('decoder_7', 1, (BF, 2, 5), 512, 512, 512, 1, (2, 2), (1, 1)),
    ('decoder_6', 1, (BF, 4, 9), 512, 512, 512, 1, (2, 2), (1, 1)),
    ('decoder_5', 1, (BF, 8, 17), 512, 512, 512, 1, (2, 2), (1, 1)),
    ('decoder_4', 1, (BF, 16, 33), 512, 512, 256, 1, (2, 2), (1, 1)),
    ('decoder_3', 1, (BF, 32, 65), 256, 256, 128, 1, (2, 2), (1, 1)),
    ('decoder_2', 1, (BF, 64, 129), 128, 128, 64, 1, (2, 2), (1, 1)),
    ('decoder_1', 1, (BF, 128, 257), 64, 64, 1, 1, (2, 2), (1, 1)),
    ('layer_1', 0, (2 * BF, 256, 513), 1, 1, 64, 0, (2, 2), (1, 1)),
    ('layer_2', 0, (2 * BF, 128, 256), 64, 0, 128, 0, (2, 2), (1, 1)),
    ('layer_3', 0, (2 * BF, 64, 128), 128, 0, 256, 0, (2, 2), (1, 1)),
    ('layer_4', 0, (2 * BF, 32, 64), 256, 0, 512, 0, (1, 1), (1, 1)),
    ('layer_5', 0, (2 * BF, 31, 63), 512, 0, 1, 0, (1, 1), (1, 1)),
]


def dot(a, b):
  return float((a.double() * b.double()).sum())


def build(case, act):
  from advoc_amd import conv
  name, kind, (n, H, W), c0, c1, cout, trim, stride, pad = case
  dev = torch.device('cuda')
  g = torch.Generator(device='cuda').manual_seed(sum(ord(ch) for ch in name))
  x0 = torch.randn(n, H, W + trim, c0, device=dev, generator=g)
  x1 = torch.randn(n, H, W, c1, device=dev, generator=g) if c1 else None
  if kind == 0:
    oh = (H + 2 * pad[0] - 4) // stride[0] + 1 if name.startswith('layer') else -(-H // stride[0])
    ow = (W + 2 * pad[1] - 4) // stride[1] + 1 if name.startswith('layer') else -(-W // stride[1])
    w = torch.randn(4, 4, c0 + c1, cout, device=dev, generator=g) * 0.05
    clip = 0
  else:
    clip = 1 if cout == 1 else 0
    oh, ow = 2 * H, 2 * W - clip
    w = torch.randn(4, 4, cout, c0 + c1, device=dev, generator=g) * 0.05
  y = torch.zeros(n, oh, ow, cout, device=dev)
  b = torch.zeros(cout, device=dev)
  L = conv.Layer(kind, x0, y, w, b, x1=x1, in_w=W, stride=stride, pad=pad, in_act=act)
  dy = torch.randn(n, oh, ow, cout, device=dev, generator=g)
  return L, x0, x1, w, b, y, dy, W


ALL_LAYERS = [('small32', c) for c in LAYERS] + [('full64', c) for c in LAYERS_FULL]


@gpu
@pytest.mark.parametrize('model,case', ALL_LAYERS, ids=['%s-%s' % (m, c[0]) for m, c in ALL_LAYERS])
def test_three_directions_are_one_trilinear_form(hip, model, case):
  L, x0, x1, w, b, y, dy, W = build(case, act=0)
  if model == 'full64' and case[0] in ('encoder_2', 'encoder_3', 'encoder_4', 'decoder_5', 'decoder_4', 'decoder_3',
                                       'layer_2', 'layer_3', 'layer_4'):
    # the launches that dominate the configs[2] step run on the operand-image kernels: the patch kernels of
    # igemm_patch.hip wherever the grid has at least 16 x 16 points that 16 x 16 patches cover with < 25 % waste (four
    # fused sub-pixel phases, the 4x4 stride-1 conv, the stride-2 gathers as parity planes), the per-tap tiles of
    # igemm_h3.hip for the rest (decoder_5 / encoder_5 gather over 8 x 17 points).  The model's 33 / 65 / 129-wide grids
    # are 16 n + 1 columns: the patches take the 16 n, a per-tap launch the last column.
    patch_fwd = case[0] in ('encoder_2', 'encoder_3', 'encoder_4', 'decoder_4', 'decoder_3', 'layer_2', 'layer_3', 'layer_4')
    patch_bwd = case[0] in ('encoder_2', 'encoder_3', 'encoder_4', 'decoder_4', 'decoder_3', 'layer_2', 'layer_3', 'layer_4')
    for d, patch in ((0, patch_fwd), (1, patch_bwd)):
      want = 'patch_gemm_h3_kernel' if patch else 'gather_gemm_h3_kernel'
      assert want in L.kernel_name(d), (d, L.kernel_name(d))
  if model == 'full64' and case[0] in ('decoder_1', 'layer_5'):
    assert 'fused_taps_kernel' in L.kernel_name(0), L.kernel_name(0)     # (r4) <= 2 output columns: one launch, S in LDS
  if model == 'full64' and case[0] == 'layer_1':
    assert 'fused_taps_kernel' in L.kernel_name(1), L.kernel_name(1)
  L.forward()
  lhs = dot(dy, y)
  dx0 = torch.zeros_like(x0)
  dx1 = torch.zeros_like(x1) if x1 is not None else None
  L.backward_data(dy, dx0, dx1)
  via_x = dot(dx0[:, :, :W], x0[:, :, :W]) + (dot(dx1, x1) if x1 is not None else 0.0)
  dw = torch.zeros_like(w)
  db = torch.zeros_like(b)
  L.backward_weight(dy, dw, db)
  via_w = dot(dw, w)
  scale = float(dy.double().norm() * y.double().norm())
  assert abs(lhs - via_x) < 2e-6 * scale, (lhs, via_x)
  assert abs(lhs - via_w) < 2e-6 * scale, (lhs, via_w)
  # bias gradient = column sums of dy
  want_db = dy.double().sum(dim=(0, 1, 2))
  assert float((db.double() - want_db).norm()) < 1e-5 * float(want_db.norm() + dy.double().norm())


HOMOG = [LAYERS[1], LAYERS[7], LAYERS[13], LAYERS_FULL[2], LAYERS_FULL[12], LAYERS_FULL[19]]


@gpu
@pytest.mark.parametrize('case', HOMOG, ids=['encoder_2', 'decoder_3', 'layer_4', 'full64-encoder_3', 'full64-decoder_4',
                                             'full64-layer_4'])
def test_leaky_relu_prologue_is_positively_homogeneous(hip, case):
  L, x0, x1, w, b, y, dy, W = build(case, act=1)
  b.copy_(torch.randn_like(b))
  L.forward()
  y1 = y.clone()
  x0.mul_(2.0)
  if x1 is not None:
    x1.mul_(2.0)
  L.forward()
  lhs, rhs = (y - b).double(), 2.0 * (y1 - b).double()
  assert float((lhs - rhs).norm() / rhs.norm()) < 1e-6


@gpu
def test_stft_istft_round_trip_full_batch(hip):
  from advoc_amd import spectral
  g = torch.Generator(device='cuda').manual_seed(5)
  n = 255 * 256 + 1024
  x = torch.rand(B, n, device='cuda', generator=g) - 0.5
  X = torch.view_as_complex(spectral._run_stft(x, 1024, 256, 256, complex_out=True))
  assert tuple(X.shape) == (B, 256, 513)
  y = spectral.istft_batch(X, 1024, 256)
  assert tuple(y.shape) == (B, n)
  assert float((y[:, 1024:-1024] - x[:, 1024:-1024]).abs().max()) < 3e-6
  # |X| from the fused magnitude kernel equals |.| of the complex kernel
  mag = spectral._run_stft(x, 1024, 256, 256, complex_out=False)
  assert float((mag - X.abs()).abs().max()) < 1e-4 * float(mag.abs().max())


@gpu
def test_dropout_stream_is_shard_invariant_at_full_size(hip):
  from advoc_amd import _lib
  lib = _lib.load()
  per_clip = 16 * 34 * 256                      # decoder_5 output of one clip
  whole = torch.empty(B * per_clip, dtype=torch.uint8, device='cuda')
  _lib.check(lib.advoc_dropout_mask_u8(_lib.ptr(whole), whole.numel(), 12345, 0, 0.5, _lib.stream()), 'mask')
  half = torch.empty(B // 2 * per_clip, dtype=torch.uint8, device='cuda')
  _lib.check(lib.advoc_dropout_mask_u8(_lib.ptr(half), half.numel(), 12345, B // 2 * per_clip, 0.5, _lib.stream()),
             'mask')
  assert torch.equal(whole[B // 2 * per_clip:], half)
  keep = float(whole.float().mean())
  assert abs(keep - 0.5) < 2e-3


@gpu
def test_offset_range_limits(hip):
  """The kernels index tensors with 32-bit element offsets.  Just under 2^31 elements the last image
  of a 2040-clip discriminator layer_1 batch (8.6 GB of output) must equal the same image run alone; past the limit the
  call is refused with ADVOC_ERR_UNSUPPORTED, not executed with wrapped offsets."""
  from advoc_amd import _lib, conv
  dev = torch.device('cuda')
  g = torch.Generator(device='cuda').manual_seed(1)
  w = torch.randn(4, 4, 2, 32, device=dev, generator=g) * 0.05
  b = torch.randn(32, device=dev, generator=g) * 0.1

  def run(n, x0, x1):
    y = torch.empty(n, 128, 256, 32, device=dev)
    conv.Layer(conv.CONV, x0, y, w, b, x1=x1, stride=(2, 2), pad=(1, 1), in_act=conv.ACT_NONE).forward()
    return y
  n = 2040                                                   # 2040*128*256*32 = 2.139e9 output elements < 2^31
  x0 = torch.randn(n, 256, 513, 1, device=dev, generator=g)
  x1 = torch.randn(n, 256, 513, 1, device=dev, generator=g)
  y = run(n, x0, x1)
  for i in (0, n // 2, n - 1):
    alone = run(1, x0[i:i + 1].contiguous(), x1[i:i + 1].contiguous())
    assert torch.equal(y[i:i + 1], alone), i
  del y
  n = 2056                                                   # 2.156e9 output elements: past int32
  y = torch.empty(n, 128, 256, 32, device=dev)
  x0 = torch.zeros(n, 256, 513, 1, device=dev)
  with pytest.raises(_lib.AdvocHipError, match='unsupported'):
    conv.Layer(conv.CONV, x0, y, w, b, x1=x0, stride=(2, 2), pad=(1, 1), in_act=conv.ACT_NONE).forward()


# layers whose launches dominate the configs[2] step (58 % patch kernels, 17 % wgrad_h3_256)
PATCH_G = ('encoder_2', 'encoder_3', 'encoder_4', 'decoder_4', 'decoder_3', 'decoder_2')
BIG_WGRAD_G = ('encoder_3', 'encoder_4', 'encoder_5', 'decoder_5', 'decoder_4', 'decoder_3', 'decoder_2')


def read_gates(st, distinct, d_passes):
  """The sign pattern of every (leaky) ReLU of the last forward passes, for the oracle's gate injection
  (oracle/advoc_torch.py: build_generator / build_discriminator `gates`): {generator layer: gate} and
  {'D/real/layer_k' | 'D/fake/layer_k': gate}, first `distinct` clips.  d_passes: [(tag, layers, {index: batch norm}, first
  clip of the pass INSIDE the layers' tensors)].  A discriminator activation that exists as the next layer's operand image
  only (conv.Layer.y_image_only: the fp32 tensor is never written) is read from that image's high plane."""
  out = {}
  names = ['encoder_%d' % (i + 1) for i in range(len(st['enc']))] + ['decoder_%d' % idx for idx in st['dec']]
  for n in names:
    k = int(n.split('_')[1])
    z = (st['enc'][k - 1] if n.startswith('enc') else st['dec'][k])[:distinct]
    b = st['g_bn'].get(n)
    gate = (z * b['scale'] + b['shift'] if b is not None else z) > 0
    if n.startswith('dec') and k in st['masks']:      # dropout behind the layer: a dropped value is 0
      gate = gate & (st['masks'][k][0][:distinct] > 0)
    out[n] = gate.cpu()
  for tag, layers, bns, lo in d_passes:
    for i in range(4):
      nxt = layers[i + 1]
      if getattr(nxt, '_x_gates', False):
        N, H, W, C = nxt.x0.shape
        h = nxt._img[0][:N * H * W * C * 2].view(torch.float16).reshape(N, H, W, C // 32, 2, 32)[:, :, :, :, 0, :]
        gate = h.reshape(N, H, W, C)[lo:lo + distinct] > 0
      else:
        z = layers[i].y[lo:lo + distinct]
        b = bns.get(i)
        gate = (z * b['scale'] + b['shift'] if b is not None else z) > 0
      out['D/%s/layer_%d' % (tag, i + 1)] = gate.cpu()
  return out, names


@gpu
def test_full_model_train_loops_at_bench_size_match_the_float64_oracle(hip):
  """BASELINE configs[2] exactly as bench.py times it -- AdVoc-full, 64 clips x 256 frames, default dispatch, delayed
  scaling, side stream -- two train_loops against the float64 oracle (models/advoc/advoc_model.py:238-257, 285-289).
  Asserts that the kernels the bench line is made of are the ones running (patch_gemm_h3_kernel on encoder_2-4 /
  decoder_2-4 / layer_2-4, wgrad_h3_256_kernel on the wide weight gradients) and that the second step builds its operand
  images in one pass (delayed scaling live).  Bars: losses 1e-4; every gradient tensor rel-L2 <= max(5e-4, 3 x what a
  float32 torch-CPU evaluation of the same graph achieves -- only evaluated for a tensor that misses 5e-4).
  The 64 clips are 8 distinct clips (and dropout masks) tiled 8 times: without batch norm every loss is a batch mean of
  per-clip terms, so the oracle evaluates the 8 distinct clips (1/8 of the float64 CPU work) while the HIP side runs the
  full 64-clip launches."""
  from advoc_amd.model import Advoc, Modes
  from oracle import advoc_torch as A
  Bn, T, DISTINCT = 64, 256, 8
  cfg = A.Config(small=False, subseq_len=T)
  P = A.init_params(cfg, seed=11)
  g = torch.Generator().manual_seed(12)
  for k in P:
    if k.endswith('/bias'):
      P[k] = torch.randn(P[k].shape, generator=g) * 0.05
  m = Advoc(Modes.TRAIN)
  m.train_batch_size = Bn
  m.build(batch_size=Bn)
  m.load_state_dict(P)
  st = m._built
  assert st['side_on'] and all(lay.delayed_scale for lay in st['g_layers'].values())

  # ---- the dispatch the bench measures ----
  GL = st['g_layers']
  for name in PATCH_G:
    assert 'patch_gemm_h3_kernel' in GL[name].kernel_name(0), (name, GL[name].kernel_name(0))
    assert 'patch_gemm_h3_kernel' in GL[name].kernel_name(1), (name, GL[name].kernel_name(1))
  for name in BIG_WGRAD_G:      # (r6: `_flat` = the instance for grid rows under 32 points, encoder_5 / decoder_5)
    want = 'wgrad_h3_256_flat_kernel' if name in ('encoder_5', 'decoder_5') else 'wgrad_h3_256_kernel'
    assert GL[name].kernel_name(2) == want, (name, GL[name].kernel_name(2))
  for layers in (st['d_layers_2b'], st['d_layers_fake']):
    for i in (1, 2, 3):
      assert 'patch_gemm_h3_kernel' in layers[i].kernel_name(0), (i, layers[i].kernel_name(0))
      assert 'patch_gemm_h3_kernel' in layers[i].kernel_name(1), (i, layers[i].kernel_name(1))
  for i in (2, 3):
    assert st['d_layers_2b'][i].kernel_name(2) == 'wgrad_h3_256_kernel'
  # the <= 2-column layers in one launch (r4: fused_taps_kernel, csrc/edge.hip)
  assert GL['decoder_1'].kernel_name(0) == 'fused_taps_kernel<1, 8>', GL['decoder_1'].kernel_name(0)
  for layers in (st['d_layers_2b'], st['d_layers_fake']):
    assert layers[4].kernel_name(0) == 'fused_taps_kernel<1, 8>', layers[4].kernel_name(0)
  assert 'fused_taps_kernel' in st['d_layers_fake'][0].kernel_name(1), st['d_layers_fake'][0].kernel_name(1)

  def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))

  def make_batch(seed):
    gg = torch.Generator().manual_seed(seed)
    target = torch.rand(DISTINCT, T, 513, 1, generator=gg) * 2
    x = target * (0.5 + torch.rand(DISTINCT, T, 513, 1, generator=gg)) - 0.1
    return x, target
  batches = [make_batch(200 + i) for i in range(4)]
  masks = [A.make_dropout_masks(cfg, DISTINCT, seed=60 + i) for i in range(4)]
  tile = lambda t: t.repeat(Bn // DISTINCT, 1, 1, 1)      # noqa: E731
  dev = torch.device('cuda')
  it = iter(range(4))

  def feed():
    i = next(it)
    m.set_dropout_masks({k: tile(v.to(torch.uint8)) for k, v in masks[i].items()})
    return tile(batches[i][0]).to(dev), tile(batches[i][1]).to(dev)
  m(feed)

  P64 = collections.OrderedDict((k, v.double()) for k, v in P.items())
  Gk, Dk = A.split_vars(P64)
  g_opt, d_opt = A.AdamTF(Gk, P64), A.AdamTF(Dk, P64)
  for step in range(2):
    bd, bg = 2 * step, 2 * step + 1
    # oracle, float64, on the 8 distinct clips
    gD, LD = A.grads_in_chunks(P64, batches[bd][0].double(), batches[bd][1].double(), cfg,
                               {k: v.double() for k, v in masks[bd].items()}, 'D', 8)
    P_at_d = collections.OrderedDict((k, v.clone()) for k, v in P64.items())
    d_opt.step(P64, gD)
    # product: the train_loop in its two halves (advoc_model.py:285-289: D update on one batch, G update on the next).  The
    # generator's gradients are differentiated at the discriminator parameters THIS run holds after its update (r5): an Adam
    # step is ~lr whatever the size of a gradient, so a discriminator parameter whose gradient is round-off on both sides can
    # move by +lr here and -lr in the oracle, and the image weight gradient sums with atomics -- compared across that
    # difference the generator's most sensitive tensor (decoder_8's kernel, fed by the 1 x 3-point bottleneck) came out
    # anywhere between 1.4e-4 and 6.6e-4 from run to run, on either side of this test's bar.  The parameters themselves are
    # compared first, then adopted.
    m.d_step(m._feed())
    gates_d, names_g = read_gates(st, DISTINCT, [('real', st['d_layers_2b'], {}, 0), ('fake', st['d_layers_2b'], {}, Bn)])
    sd = m.state_dict()
    for k in Dk:
      assert rel(sd[k], P64[k]) < 1e-4, (step, k, rel(sd[k], P64[k]))
      P64[k] = sd[k].double().cpu()
    gG, LG = A.grads_in_chunks(P64, batches[bg][0].double(), batches[bg][1].double(), cfg,
                               {k: v.double() for k, v in masks[bg].items()}, 'G', 8)
    P_at_g = collections.OrderedDict((k, v.clone()) for k, v in P64.items())
    g_opt.step(P64, gG)
    m.g_step(m._feed())
    gates_g, _ = read_gates(st, DISTINCT, [('fake', st['d_layers_fake'], {}, 0)])
    assert m.step == step + 1
    # (r6, VERDICT r5 item 5a / ADVICE r5) the read-back sign patterns -- the generator's from its fp32 pre-activations, the
    # discriminator's partly from the fp16 high plane of an operand IMAGE (y_image_only) -- against the float64 oracle's OWN
    # forward passes at the same parameters: a kernel (or an image-sign path) that computes a wrong gate cannot hide behind the
    # gate-frozen comparison below, whose gates are this run's.  Up to 1e-4 of the gates may differ (pre-activations within
    # round-off of zero).  And the north-star's literal bar at the BENCHED dispatch: generator output within 1e-4 rel-L2.
    for tagp, P_at, bi, gts, passes in (('D', P_at_d, bd, gates_d, ('real', 'fake')), ('G', P_at_g, bg, gates_g, ('fake',))):
      xb, tb = batches[bi][0].double(), batches[bi][1].double()
      col = []
      with torch.no_grad():
        gen64 = A.build_generator(P_at, xb, cfg, {k: v.double() for k, v in masks[bi].items()}, collect=col)
        flips = tot = 0
        for n, c in zip(names_g, col):
          flips += int((gts[n] != (c > 0)).sum())
          tot += c.numel()
        for tag in passes:
          acts = []
          A.build_discriminator(P_at, xb, tb if tag == 'real' else gen64, cfg, collect=acts)
          for i in range(4):
            flips += int((gts['D/%s/layer_%d' % (tag, i + 1)] != (acts[i] > 0)).sum())
            tot += acts[i].numel()
      print('step %d %s pass: gates that differ from the float64 forward passes: %d of %d' % (step + 1, tagp, flips, tot))
      assert flips <= 1e-4 * tot, (tagp, flips, tot)
      if tagp == 'G':
        r_gen = rel(st['gen_out'][:DISTINCT], gen64)
        print('step %d: generator output at the 64-clip dispatch vs float64: rel-L2 %.3g' % (step + 1, r_gen))
        assert r_gen < 1e-4, r_gen
        # (the last of the 8 tilings against the first: the same clips through other tiles / K-slice orders of the same launches)
        assert rel(st['gen_out'][Bn - DISTINCT:], gen64) < 1e-4 and rel(st['gen_out'][Bn - DISTINCT:], st['gen_out'][:DISTINCT]) < 1e-5
    # GATE-FROZEN (r5): the same gradients against the float64 oracle evaluated with THIS run's sign patterns (every leaky /
    # plain ReLU of the generator and of the discriminator's passes): a smooth function on both sides, every tensor held to
    # 5e-4 with no recourse to what float32 achieves.  The free-running comparison below keeps its r3 form; its one sensitive
    # tensor, decoder_8's kernel behind the 1 x 3-point bottleneck, lands between 6e-5 and 7e-4 from run to run depending on
    # which side of zero a handful of bottleneck pre-activations fall.
    fD, _ = A.grads_in_chunks(P_at_d, batches[bd][0].double(), batches[bd][1].double(), cfg,
                              {k: v.double() for k, v in masks[bd].items()}, 'D', 8, gates=gates_d)
    fG, _ = A.grads_in_chunks(P_at_g, batches[bg][0].double(), batches[bg][1].double(), cfg,
                              {k: v.double() for k, v in masks[bg].items()}, 'G', 8, gates=gates_g)
    frozen = {}
    for net, want in (('d_G', fD), ('g_G', fG)):
      for k, v in want.items():
        frozen[k] = rel(st[net][k], v)
    print('step %d: gate-frozen oracle: worst gradient rel-L2 vs float64 %.3g (%s)'
          % (step + 1, max(frozen.values()), max(frozen, key=frozen.get)))
    assert all(r <= 5e-4 for r in frozen.values()), sorted(frozen.items(), key=lambda kv: -kv[1])[:4]
    ls = m.losses()
    assert abs(ls['disc_loss'] - LD['d_loss']) < 1e-4 * max(1, abs(LD['d_loss'])), (step, ls, LD)
    assert abs(ls['gen_loss_GAN'] - LG['g_gan']) < 1e-4 * max(1, abs(LG['g_gan'])), (step, ls, LG)
    assert abs(ls['gen_loss_L1'] - LG['g_l1']) < 1e-4 * max(1, abs(LG['g_l1'])), (step, ls, LG)
    # FREE-RUNNING: the oracle with its own gates.  A report with a FIXED bar since r5 (2e-3; r3 / r4: 5e-4 or three times what a
    # float32 torch-CPU evaluation achieved on the tensor -- a bar that followed the observation, and flaky: the same build gave
    # 6e-5 .. 7e-4 on decoder_8's kernel from run to run, each side of it).  What it can catch that the gate-frozen comparison
    # cannot is a WRONG gate, and a wrong gate anywhere but at a pre-activation within round-off of zero is far beyond 2e-3.
    worst = {}
    for net, want in (('d_G', gD), ('g_G', gG)):
      for k, v in want.items():
        worst[k] = rel(st[net][k], v)
    print('step %d: worst gradient rel-L2 vs float64 %.3g (%s)' % (step + 1, max(worst.values()), max(worst, key=worst.get)))
    assert all(r <= 2e-3 for r in worst.values()), sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    # delayed scaling is live from the second step on: every persistent image header holds a previous magnitude
    if step == 1:
      hdrs = [h for lay in list(GL.values()) + st['d_layers_2b'] + st['d_layers_fake'] for h in lay.image_headers()]
      # (r5: or, for an output-gradient image the layer above writes under its a-priori scale -- no history by design --, the
      # magnitude it recorded and its 2^-s: words 0 and 1)
      assert hdrs and all(int(h[2]) != 0 or (int(h[0]) != 0 and int(h[1]) != 0) for h in torch.stack(hdrs).cpu())
    # the next step starts from IDENTICAL parameters on both sides (Adam's first steps turn round-off in near-zero
    # gradients into +-lr differences; the updates themselves are compared in test_hip_model.py)
    sd = m.state_dict()
    for k in P64:
      assert rel(sd[k], P64[k]) < 1e-4, (step, k, rel(sd[k], P64[k]))
      P64[k] = sd[k].double().cpu()
  print('image refits over two steps: %d' % m.image_refits())


@gpu
def test_full_model_with_batch_norm_train_loop_at_bench_size_matches_the_float64_oracle(hip):
  """use_batchnorm=True at BASELINE configs[2]'s size (AdVoc-full, 64 clips x 256 frames, default dispatch, side stream):
  one train_loop -- D update (two D passes, real and fake, each with its own batch statistics: advoc_model.py:168-204 is
  built twice) and G update -- against the float64 oracle (advoc_model.py:77-84,173-177: tf.layers.batch_normalization,
  training=True).  This is the configuration the r3 side-stream corruption lived in (NOTEBOOK.md section 5), now with the
  inter-pass join removed.  The 64 clips are 8 distinct clips tiled 8 times: tiling a batch changes neither its per-channel
  mean nor its variance, and every loss is a batch mean, so the oracle evaluates the 8 distinct clips in ONE call (batch
  norm needs the whole batch at once) while the HIP side runs the full 64-clip launches.  Bars as in the test above."""
  from advoc_amd.model import Advoc, Modes
  from oracle import advoc_torch as A
  Bn, T, DISTINCT = 64, 256, 8
  cfg = A.Config(small=False, subseq_len=T, use_batchnorm=True)
  P = A.init_params(cfg, seed=21)
  g = torch.Generator().manual_seed(22)
  for k in P:
    if k.endswith('/bias') or k.endswith('/beta'):
      P[k] = torch.randn(P[k].shape, generator=g) * 0.05
    elif k.endswith('/gamma'):
      P[k] = 1.0 + torch.randn(P[k].shape, generator=g) * 0.1
  m = Advoc(Modes.TRAIN)
  m.train_batch_size = Bn
  m.use_batchnorm = True
  m.build(batch_size=Bn)
  m.load_state_dict(P)
  st = m._built
  assert st['side_on'] and st['bn_on']
  GL = st['g_layers']
  for name in PATCH_G:
    assert 'patch_gemm_h3_kernel' in GL[name].kernel_name(0), (name, GL[name].kernel_name(0))
    assert 'patch_gemm_h3_kernel' in GL[name].kernel_name(1), (name, GL[name].kernel_name(1))
  for layers in (st['d_layers_real'], st['d_layers_fake']):
    for i in (1, 2, 3):
      assert 'patch_gemm_h3_kernel' in layers[i].kernel_name(1), (i, layers[i].kernel_name(1))
  # (with batch norm the inputs of these layers carry the producer's affine: fused_taps_kernel applies it before the zero padding)
  assert GL['decoder_1'].kernel_name(0) == 'fused_taps_kernel<1, 8>', GL['decoder_1'].kernel_name(0)
  assert st['d_layers_fake'][4].kernel_name(0) == 'fused_taps_kernel<1, 8>', st['d_layers_fake'][4].kernel_name(0)

  def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))

  def make_batch(seed):
    gg = torch.Generator().manual_seed(seed)
    target = torch.rand(DISTINCT, T, 513, 1, generator=gg) * 2
    x = target * (0.5 + torch.rand(DISTINCT, T, 513, 1, generator=gg)) - 0.1
    return x, target
  batches = [make_batch(300 + i) for i in range(2)]
  masks = [A.make_dropout_masks(cfg, DISTINCT, seed=80 + i) for i in range(2)]
  tile = lambda t: t.repeat(Bn // DISTINCT, 1, 1, 1)      # noqa: E731
  dev = torch.device('cuda')
  it = iter(range(2))

  def feed():
    i = next(it)
    m.set_dropout_masks({k: tile(v.to(torch.uint8)) for k, v in masks[i].items()})
    return tile(batches[i][0]).to(dev), tile(batches[i][1]).to(dev)
  m(feed)

  P64 = collections.OrderedDict((k, v.double()) for k, v in P.items())
  Gk, Dk = A.split_vars(P64)
  d_opt = A.AdamTF(Dk, P64)
  gD, LD = A.grads(P64, batches[0][0].double(), batches[0][1].double(), cfg, {k: v.double() for k, v in masks[0].items()}, 'D')
  P_at_d = collections.OrderedDict((k, v.clone()) for k, v in P64.items())
  d_opt.step(P64, gD)
  # the train_loop, in its two halves, so that the generator's sign patterns of BOTH forward passes can be read back
  names = ['encoder_%d' % (i + 1) for i in range(len(st['enc']))] + ['decoder_%d' % idx for idx in st['dec']]

  def read_gates():
    """{layer: (value its consumers put through their (leaky) ReLU) > 0} of the forward pass just run, distinct clips only"""
    out = {}
    for n in names:
      z = (st['enc'][int(n.split('_')[1]) - 1] if n.startswith('enc') else st['dec'][int(n.split('_')[1])])[:DISTINCT]
      b = st['g_bn'].get(n)
      v = z * b['scale'] + b['shift'] if b is not None else z
      gate = v > 0
      if n.startswith('dec') and int(n.split('_')[1]) in st['masks']:      # dropout behind the batch norm: a dropped value is 0
        gate = gate & (st['masks'][int(n.split('_')[1])][0][:DISTINCT] > 0)
      out[n] = gate.cpu()
    return out

  def read_d_gates(out, layers_bns):
    """+ the discriminator's leaky-ReLU sign patterns: {'D/real/layer_k' | 'D/fake/layer_k': gate}, k = 1 .. 4"""
    for tag, bns, lo in layers_bns:
      for i in range(4):
        z = st['d_act'][i][lo:lo + DISTINCT]
        b = bns.get(i)
        out['D/%s/layer_%d' % (tag, i + 1)] = ((z * b['scale'] + b['shift'] if b is not None else z) > 0).cpu()
    return out
  b0 = m._feed()
  m.d_step(b0)
  gates_d = read_d_gates(read_gates(), (('real', st['d_bn_real'], 0), ('fake', st['d_bn_fake'], Bn)))
  # The generator update is differentiated at the parameters THIS run holds after its discriminator update (r5).  Adam's
  # first step is +-lr for every parameter whatever the size of its gradient, so a discriminator parameter whose gradient is
  # round-off on both sides (e.g. a conv bias in front of a batch norm) moves by +lr here and -lr in the oracle; r4 compared
  # the generator's gradients across that difference and read the result as ill-conditioning of the bottleneck.
  sd = m.state_dict()
  d_dev = max(rel(sd[k], P64[k]) for k in Dk)
  print('discriminator parameters after the update vs the oracle\'s own update: worst rel-L2 %.3g' % d_dev)
  P_at_g = collections.OrderedDict((k, sd[k].double().cpu()) for k in P64)
  gG, LG = A.grads(P_at_g, batches[1][0].double(), batches[1][1].double(), cfg, {k: v.double() for k, v in masks[1].items()}, 'G')
  b1 = m._feed()
  m.g_step(b1)
  gates_g = read_d_gates(read_gates(), (('fake', st['d_bn_fake'], Bn),))
  assert m.step == 1
  ls = m.losses()
  assert abs(ls['disc_loss'] - float(LD['d_loss'])) < 1e-4 * max(1, abs(float(LD['d_loss']))), (ls, LD)
  assert abs(ls['gen_loss_GAN'] - float(LG['g_gan'])) < 1e-4 * max(1, abs(float(LG['g_gan']))), (ls, LG)
  assert abs(ls['gen_loss_L1'] - float(LG['g_l1'])) < 1e-4 * max(1, abs(float(LG['g_l1']))), (ls, LG)

  def compare(want_d, want_g):
    worst = {}
    for net, want in (('d_G', want_d), ('g_G', want_g)):
      for k, v in want.items():
        if float(v.norm()) < 1e-9 * (1 + v.numel()) ** 0.5:
          continue      # a conv bias in front of a batch norm: its gradient is exactly zero, both sides are round-off
        worst[k] = rel(st[net][k], v)
    return worst
  # (1) END TO END against the free-running float64 oracle.  r3 / r4 saw 1.0e-2 .. 2.7e-2 here on the tensors behind the
  # 1 x 3-point bottleneck, in two "modes" that followed the order of a few fp32 additions, explained it with ReLU gates
  # flipping on round-off and held it to a multiple of float32's own distance -- a bar that moved with the observation.  r5
  # measured the explanation: 25 of 65 million generator gates differ from the float64 forward pass, and freezing them changes
  # NOTHING (2.6e-2 either way); differentiating the generator at the discriminator parameters this run actually holds after
  # its update (above) takes the same comparison to 2.9e-3 .. 3.2e-3 -- the distance a float32 torch-CPU evaluation of the
  # graph has from float64 (3.5e-3).  The two modes were the +-lr of Adam's first step on parameters with round-off
  # gradients.  Fixed bars: generator 1e-2, discriminator 2e-3; the kernels' precision is pinned by (2).
  worst = compare(gD, gG)
  top = sorted(worst.items(), key=lambda kv: -kv[1])[:6]
  print('batch norm, one train_loop at 64 x 256, free-running oracle: worst gradients rel-L2 vs float64: %s'
        % ', '.join('%s %.3g' % kv for kv in top))
  assert all(r <= (2e-3 if k.startswith('discriminator/') else 1e-2) for k, r in worst.items()), top
  # (2) GATE-FROZEN: the float64 oracle evaluated with the sign pattern of every (leaky) ReLU of the
  # generator AND of the discriminator's passes (measured: generator gates alone 1.5e-3, with the discriminator's 4.0e-4)
  # taken from THIS run's forward passes (read back above: pre-activation tensors and the batch-norm affines the
  # kernels computed) instead of from its own values.  What is compared is then a smooth function of the parameters on both
  # sides -- no gate can be decided by a round-off error -- and every gradient tensor, the ones behind the 1 x 3 bottleneck
  # included, is held to 1e-3 (the model without batch norm: 5e-4, test above).  A kernel that lost precision behind the bottleneck
  # (the fp16-pair images, the K-slice sums, the batch-norm reductions) fails here, however the gates fall.
  flips = tot = 0
  col = []
  A.build_generator(P_at_g, batches[1][0].double(), cfg, {k: v.double() for k, v in masks[1].items()}, collect=col)
  for n, c in zip(names, col):
    flips += int((gates_g[n] != (c > 0)).sum())
    tot += c.numel()
  print('gates that differ from the float64 forward pass: %d of %d' % (flips, tot))
  assert flips <= 1e-4 * tot, (flips, tot)        # the read-back gates ARE the forward pass's, up to round-off ties
  fD, _ = A.grads(P_at_d, batches[0][0].double(), batches[0][1].double(), cfg, {k: v.double() for k, v in masks[0].items()},
                  'D', gates=gates_d)
  fG, _ = A.grads(P_at_g, batches[1][0].double(), batches[1][1].double(), cfg, {k: v.double() for k, v in masks[1].items()},
                  'G', gates=gates_g)
  frozen = compare(fD, fG)
  top = sorted(frozen.items(), key=lambda kv: -kv[1])[:6]
  print('gate-frozen oracle: worst gradients rel-L2 vs float64: %s' % ', '.join('%s %.3g' % kv for kv in top))
  # (bar: 7e-4, a fixed number -- measured 4.0e-4 .. 6.2e-4 from run to run, uniformly over the generator's tensors: what is
  # left with every gate frozen is fp32 round-off through batch statistics over 24 samples; float32 torch-CPU's own distance on
  # this graph is 3.5e-3, r4 accepted 2.7e-2 here)
  # (r6: 7e-4 -- VERDICT r5 item 5b)
  assert all(r <= 7e-4 for r in frozen.values()), top
