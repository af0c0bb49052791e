"""AdVoc / AdVoc-small networks, losses, gradients and one full train_loop on MI355X against
the torch-CPU oracle (same weights, same inputs, injected dropout masks).
Bar: 1e-4 relative L2 (north_star); fp32-exact MFMA typically lands at ~1e-6."""
import numpy as np
import pytest
import torch

from oracle import advoc_torch as A

gpu = pytest.mark.gpu
BAR = 1e-4        # north_star bar, applied to every forward tensor (achieved: ~1e-6)
GRAD_BAR = 5e-4   # gradients through up to 37 stacked fp32 layers (G step: G fwd, D fwd, D bwd, G bwd);
                  # the oracle side is float64, and a float32 torch-CPU run of the same graph is
                  # measured alongside to show this is fp32 round-off, not a defect


def rel(a, b):
  a = a.detach().double().cpu()
  b = b.detach().double().cpu()
  return float((a - b).norm() / b.norm().clamp_min(1e-30))


def close(a, b, bar):
  """relative L2 below `bar`, or both negligible: a conv bias followed by batch norm has an
  exactly-zero gradient, where the fp32 result is pure round-off (~1e-8) and a ratio is meaningless."""
  a = a.detach().double().cpu()
  b = b.detach().double().cpu()
  return float((a - b).norm()) < max(bar * float(b.norm()), 1e-6 * (1 + a.numel()) ** 0.5 * 1e-1)


def make(small, subseq_len, B, seed=0, bn=False, ngf=None, ndf=None):
  from advoc_amd.model import Advoc, AdvocSmall, Modes
  kw = {}
  if ngf is not None:
    kw['ngf'] = ngf
  if ndf is not None:
    kw['ndf'] = ndf
  cfg = A.Config(small=small, subseq_len=subseq_len, use_batchnorm=bn, **kw)
  P = A.init_params(cfg, seed=seed)
  # non-zero biases (and non-trivial BN affine) so those paths are exercised
  g = torch.Generator().manual_seed(seed + 1)
  for k in P:
    if k.endswith('/bias') or k.endswith('/beta'):
      P[k] = torch.randn(P[k].shape, generator=g) * 0.05
    elif k.endswith('/gamma'):
      P[k] = 1.0 + torch.randn(P[k].shape, generator=g) * 0.1
  m = (AdvocSmall if small else Advoc)(Modes.TRAIN)
  m.use_batchnorm = bn
  if ngf is not None:
    m.ngf = ngf
  if ndf is not None:
    m.ndf = ndf
  m.subseq_len = subseq_len
  m.train_batch_size = B
  m.build(batch_size=B)
  m.load_state_dict(P)
  return cfg, P, m


def batch(B, T, seed):
  g = torch.Generator().manual_seed(seed)
  target = torch.rand(B, T, 513, 1, generator=g) * 2
  x = target * (0.5 + torch.rand(B, T, 513, 1, generator=g)) - 0.1
  return x, target


def dev_masks(masks):
  return {k: v.to(torch.uint8) for k, v in masks.items()}


@gpu
@pytest.mark.parametrize('small,T,B,bn', [(True, 32, 2, False), (True, 64, 1, False), (False, 256, 1, False),
                                          (True, 32, 2, True), (True, 64, 3, True),
                                          # (1,2)-stride layers: encoder_7/8 + decoder_8/7 of the full model at 64
                                          # frames (the SC09 setting, models/advoc/melspecVocoder.py:114)
                                          (False, 64, 1, False), (False, 64, 2, True)])
def test_forward_losses_and_gradients(hip, small, T, B, bn):
  cfg, P, m = make(small, T, B, bn=bn)
  x, target = batch(B, T, 5)
  masks = A.make_dropout_masks(cfg, B, seed=3)
  m.set_dropout_masks(dev_masks(masks))
  dev = torch.device('cuda')
  xb, tb = x.to(dev), target.to(dev)

  # generator forward, every layer
  coll = []
  gen_o = A.build_generator(P, x, cfg, masks, coll)
  gen = m.build_generator(xb)
  st = m._built
  if not bn:      # with BN the stored tensors are the raw conv outputs (the normalised ones never exist)
    for i, e in enumerate(st['enc']):
      assert rel(e, coll[i]) < BAR, ('encoder', i + 1, rel(e, coll[i]))
  assert rel(gen, gen_o) < BAR
  assert tuple(gen.shape) == (B, T, 513, 1)

  # discriminator probabilities
  p_o = A.build_discriminator(P, x, target, cfg)
  p = m.build_discriminator(xb, tb)
  assert tuple(p.shape) == tuple(p_o.shape) and rel(p, p_o) < BAR

  # D step gradients + loss (float64 oracle)
  P64 = {k: v.double() for k, v in P.items()}
  m64 = {k: v.double() for k, v in masks.items()}
  gD, LD = A.grads(P64, x.double(), target.double(), cfg, m64, 'D')
  m._allreduce = None
  lr = m._lr
  m._lr = 0.0           # freeze parameters: inspect raw gradients
  m.d_step((xb, tb))
  assert abs(m.losses()['disc_loss'] - float(LD['d_loss'])) < 1e-4 * max(1, abs(float(LD['d_loss'])))
  gD32, _ = A.grads(P, x, target, cfg, masks, 'D')
  for k, v in gD.items():
    # judged like the G gradients below: against the bar, or against what plain fp32 evaluation of the
    # same graph achieves where a leaky-ReLU gate sits within round-off of 0 (batch-norm cases)
    assert close(st['d_G'][k], v, max(GRAD_BAR, 3 * rel(gD32[k], v))), (k, rel(st['d_G'][k], v), rel(gD32[k], v))

  # G step gradients + losses
  gG, LG = A.grads(P64, x.double(), target.double(), cfg, m64, 'G')
  gG32, _ = A.grads(P, x, target, cfg, masks, 'G')
  m.g_step((xb, tb))
  ls = m.losses()
  assert abs(ls['gen_loss_GAN'] - float(LG['g_gan'])) < 1e-4 * max(1, abs(float(LG['g_gan'])))
  assert abs(ls['gen_loss_L1'] - float(LG['g_l1'])) < 1e-4 * max(1, abs(float(LG['g_l1'])))
  assert abs(ls['gen_loss_total'] - float(LG['g_loss'])) < 1e-4 * max(1, abs(float(LG['g_loss'])))
  keys = [k for k, v in gG.items() if float(v.norm()) > 1e-9]     # skip exactly-zero gradients (bias before BN)
  worst = max(rel(st['g_G'][k], gG[k]) for k in keys)
  worst32 = max(rel(gG32[k], gG[k]) for k in keys)
  print('worst G-grad rel-L2 vs float64 oracle: HIP %.3g, torch-CPU float32 %.3g' % (worst, worst32))
  for k, v in gG.items():
    # the 16-layer full model is ill-conditioned in fp32 at its 1x3 bottleneck (ReLU gates flip on
    # round-off): judge each tensor against what plain fp32 evaluation of the same graph achieves
    assert close(st['g_G'][k], v, max(GRAD_BAR, 3 * rel(gG32[k], v))), (k, rel(st['g_G'][k], v), rel(gG32[k], v))
  assert worst < 4 * max(worst32, 2e-5)     # no worse than ordinary fp32 evaluation of the same graph
  m._lr = lr


@gpu
@pytest.mark.parametrize('bn', [False, True])
def test_train_loop_matches_oracle_trainer(hip, bn):
  """Full train_loop iterations (D on batch k, G on batch k+1, TF Adam) vs the oracle: two without
  batch norm, one with it (see the conditioning note below)."""
  small, T, B = True, 32, 2
  cfg, P, m = make(small, T, B, seed=7, bn=bn)
  tr = A.Trainer(cfg, seed=7)
  tr.P = {k: v.clone() for k, v in P.items()}
  Gk, Dk = A.split_vars(tr.P)
  tr.g_opt, tr.d_opt = A.AdamTF(Gk, tr.P), A.AdamTF(Dk, tr.P)
  dev = torch.device('cuda')
  batches = [batch(B, T, 100 + i) for i in range(4)]
  masks = [A.make_dropout_masks(cfg, B, seed=50 + i) for i in range(4)]
  it = iter(range(4))

  def feed():
    i = next(it)
    m.set_dropout_masks(dev_masks(masks[i]))
    return batches[i][0].to(dev), batches[i][1].to(dev)
  m(feed)
  for step in range(1 if bn else 2):
    s_o, info = tr.train_loop(batches[2 * step], batches[2 * step + 1], masks[2 * step], masks[2 * step + 1])
    s = m.train_loop()
    assert s == s_o == step + 1
    ls = m.losses()
    assert abs(ls['disc_loss'] - info['d_loss']) < 1e-4 * max(1, abs(info['d_loss']))
    assert abs(ls['gen_loss_total'] - info['g_loss']) < 1e-4 * max(1, abs(info['g_loss']))
  sd = m.state_dict()
  for k, v in tr.P.items():
    # Adam's first steps move every weight by ~lr regardless of gradient size: compare the UPDATE
    upd_o = v - P[k]
    upd = sd[k].cpu() - P[k]
    if not bn:
      assert rel(upd, upd_o) < 2e-3, (k, rel(upd, upd_o))
    elif not (k.endswith('/bias') and 'decoder_1' not in k and 'encoder_1' not in k
              and 'layer_1' not in k and 'layer_5' not in k):
      # (a bias in front of a batch norm has a zero gradient: Adam turns its round-off into +-lr steps)
      # With batch norm at this tiny size the comparison is ill-conditioned: a pre-activation that
      # lands within fp32 round-off of 0 flips a leaky-ReLU gate (measured: one flip in
      # discriminator layer_4 moves that gradient by 0.8 %), and Adam's first steps turn every
      # near-zero gradient element whose sign changes into a full +-lr difference.  Any correct fp32
      # evaluation order shows this (5 of 6 seeds tried) and a second iteration compounds it, so
      # ONE iteration is compared and the bar is on the FRACTION of weights that moved differently; gradient accuracy itself is pinned by
      # test_forward_losses_and_gradients.
      off = int(((upd - upd_o).abs() > 0.25 * upd_o.abs().max()).sum())
      assert off <= max(2, 0.05 * upd.numel()), (k, off, upd.numel(), rel(upd, upd_o))
    assert rel(sd[k], v) < 1e-5 or float((sd[k].cpu() - v).abs().max()) < 1e-3, k


@gpu
def test_reference_train_schedule_consumes_two_batches(hip):
  """train_loop pulls TWO batches per iteration when gan_weight > 0, one otherwise
  (advoc_model.py:285-289)."""
  cfg, P, m = make(True, 32, 1)
  dev = torch.device('cuda')
  calls = []

  def feed():
    calls.append(1)
    x, t = batch(1, 32, len(calls))
    return x.to(dev), t.to(dev)
  m(feed)
  m.train_loop()
  assert len(calls) == 2
  m.gan_weight = 0.
  m.train_loop()
  assert len(calls) == 3 and m.step == 2
  d_before = m.state_dict()
  m.train_loop()
  d_after = m.state_dict()
  # with gan_weight <= 0 the discriminator is not updated
  for k in d_before:
    if k.startswith('discriminator'):
      assert torch.equal(d_before[k], d_after[k])


@gpu
def test_dropout_stream_statistics_and_determinism(hip):
  from advoc_amd import _lib
  dev = torch.device('cuda')
  n = 1 << 20
  a = torch.zeros(n, dtype=torch.uint8, device=dev)
  b = torch.zeros(n, dtype=torch.uint8, device=dev)
  lib = _lib.load()
  for keep in (0.5, 0.8):
    _lib.check(lib.advoc_dropout_mask_u8(_lib.ptr(a), n, 1234, 0, keep, _lib.stream()))
    _lib.check(lib.advoc_dropout_mask_u8(_lib.ptr(b), n, 1234, 0, keep, _lib.stream()))
    assert torch.equal(a, b)
    assert abs(float(a.float().mean()) - keep) < 3e-3
    assert set(a.unique().tolist()) <= {0, 1}
  # sharding invariance: the second half drawn with an offset equals the tail of the full draw
  _lib.check(lib.advoc_dropout_mask_u8(_lib.ptr(a), n, 99, 0, 0.5, _lib.stream()))
  half = torch.zeros(n // 2, dtype=torch.uint8, device=dev)
  _lib.check(lib.advoc_dropout_mask_u8(_lib.ptr(half), n // 2, 99, n // 2, 0.5, _lib.stream()))
  assert torch.equal(half, a[n // 2:])
  _lib.check(lib.advoc_dropout_mask_u8(_lib.ptr(b), n, 100, 0, 0.5, _lib.stream()))
  assert not torch.equal(a, b)


@gpu
def test_adam_matches_tf_formula(hip):
  from advoc_amd import _lib
  import math
  dev = torch.device('cuda')
  g = torch.Generator().manual_seed(0)
  n = 1000 + 3
  p0 = torch.zeros(n)   # start at 0 so the update itself is what is compared (no cancellation)
  grads = [torch.randn(n, generator=g) * 10 ** (-i) for i in range(3)]
  p = p0.clone().to(dev)
  pad = lambda t: t  # noqa: E731
  m_ = torch.zeros(n, device=dev)
  v_ = torch.zeros(n, device=dev)
  pr, mr, vr = p0.double().clone(), torch.zeros(n).double(), torch.zeros(n).double()
  for t, gr in enumerate(grads, 1):
    lr_t = 0.0002 * math.sqrt(1 - 0.999 ** t) / (1 - 0.5 ** t)
    gr_d = gr.to(dev)
    _lib.check(_lib.load().advoc_adam_tf_f32(_lib.ptr(p), _lib.ptr(gr_d), _lib.ptr(m_), _lib.ptr(v_),
                                             n, lr_t, 0.5, 0.999, 1e-8, 0.5, _lib.stream()))
    gd = gr.double() * 0.5
    mr = 0.5 * mr + 0.5 * gd
    vr = 0.999 * vr + 0.001 * gd * gd
    pr = pr - lr_t * mr / (vr.sqrt() + 1e-8)
  assert rel(p.cpu() - p0, (pr - p0.double())) < 1e-5


@gpu
def test_short_training_run_reduces_l1(hip):
  """80 train_loop iterations on one fixed synthetic batch: every loss stays finite and the L1 term
  (weight 10 of the generator objective, advoc_model.py:243-245) falls well below its starting value.
  A sanity check of the whole optimisation loop over many steps, not a parity test."""
  from advoc_amd.model import AdvocSmall, Modes
  m = AdvocSmall(Modes.TRAIN)
  m.subseq_len = 32
  m.train_batch_size = 4
  m.build(batch_size=4, seed=3)
  x, target = batch(4, 32, 77)
  dev = torch.device('cuda')
  m((x.to(dev), target.to(dev)))
  m.train_loop()
  first = m.losses()
  hist = []
  for _ in range(80):
    m.train_loop()
    ls = m.losses()
    assert all(np.isfinite(v) for v in ls.values()), ls
    hist.append(ls['gen_loss_L1'])
  assert m.step == 81
  assert np.mean(hist[-5:]) < 0.6 * first['gen_loss_L1'], (first, hist[-5:])


@gpu
@pytest.mark.parametrize('bn', [False, True])
def test_side_stream_weight_gradients_equal_serial_execution(hip, monkeypatch, bn):
  """Weight / bias gradients run on a side stream by default (model._wgrad_ctx); ADVOC_WGRAD_STREAM=0 keeps
  the step on one stream.  Same gradients (up to the order of the weight-gradient atomics), same losses
  after a few steps -- a missing stream dependency would show as stale or torn gradients.  bn=True: the D step makes two
  passes, so the layer_1 weight gradient of pass 0 (side stream, bias sums in the layer's own table) overlaps the
  backward-data calls of pass 1 (main stream, shared workspace)."""
  from advoc_amd.model import AdvocSmall, Modes
  dev = torch.device('cuda')
  x, target = batch(16, 128, 9)
  x, target = x.to(dev), target.to(dev)

  def run(side):
    monkeypatch.setenv('ADVOC_WGRAD_STREAM', '1' if side else '0')
    m = AdvocSmall(Modes.TRAIN)
    m.subseq_len = 128
    m.train_batch_size = 16
    m.use_batchnorm = bn
    m.build(batch_size=16, seed=4)
    assert m._built['side_on'] == side
    m((x, target))
    m.train_loop()
    torch.cuda.synchronize()
    st = m._built
    grads = {k: v.detach().clone() for name in ('d_G', 'g_G') for k, v in st[name].items()}
    for _ in range(5):
      m.train_loop()
    return grads, m.losses()

  for trial in range(2):
    g1, l1 = run(True)
    g0, l0 = run(False)
    for k in g0:
      if k.startswith('discriminator') and (not bn or k.endswith('layer_1/conv2d/bias') or k.endswith('layer_1/conv2d/kernel')
                                            or k.endswith('layer_5/conv2d/bias')):          # taken at identical weights
        # The real and the fake half of the 2B batch pull these gradients in opposite directions and nearly cancel at
        # initialisation, which amplifies the order of the fp32 atomics (the only thing that differs between the two
        # schedules) ~100x: measured 2e-7 .. 9e-6 between two runs of the SAME schedule (tools/micro/side_race.py);
        # a missing dependency shows as >= 1e-2.
        # (bn=True: the biases in front of a batch norm have exactly-zero gradients, pure round-off on both sides)
        assert rel(g1[k], g0[k]) < 5e-5, (trial, bn, k, rel(g1[k], g0[k]))
    for key in ('gen_loss_L1', 'disc_loss', 'gen_loss_GAN'):
      assert abs(l1[key] - l0[key]) <= 0.02 * abs(l0[key]) + 1e-3, (trial, key, l1[key], l0[key])


@gpu
@pytest.mark.parametrize('small,bn', [(True, False), (True, True), (False, False), (False, True)])
def test_side_stream_step_equals_serial_step_element_by_element(hip, monkeypatch, small, bn):
  """The r3 "side-stream race" regression (NOTEBOOK.md section 5; csrc/lds_dma.h, dma_ring_barrier): with the weight gradients
  on the side stream a D step + G step must produce what the one-stream schedule produces -- EVERY backward-data output
  (discriminator, encoder and decoder gradients) BIT FOR BIT, every generator / discriminator gradient element-wise within
  the order-of-atomics noise of the kernels that still add with atomics (measured <= 3e-7 of the tensor's largest element;
  the race gave 1e-3) -- 20 trials per configuration, batch norm off and on (on: two D passes, pass 0's weight gradients run
  beside pass 1's backward-data launches, the overlap r3 had to forbid), on AdVoc-small and on AdVoc-full at 16 clips x 128
  frames, where the full model dispatches the patch kernels (encoder_2-4 / decoder_2-4 / layer_2-3), the per-tap image
  kernel, its workspace K split for the deep layers and remainder columns, and both weight-gradient tiles.  Two train
  steps per trial, the second from the same parameters on one-pass (delayed-scale) and producer-written images.  The image weight
  gradient sums its K slices in order (ADVOC_WGRAD_H3_ORDERED=2) so that it is reproducible on both schedules."""
  from advoc_amd import _lib
  from advoc_amd.model import Advoc, AdvocSmall, Modes
  dev = torch.device('cuda')
  B, T = 16, 128
  x, target = batch(B, T, 9)
  x, target = x.to(dev), target.to(dev)
  monkeypatch.setenv('ADVOC_WGRAD_H3_ORDERED', '2')
  _lib.reload_env()
  names = {}

  def run(side):
    monkeypatch.setenv('ADVOC_WGRAD_STREAM', '1' if side else '0')
    m = (AdvocSmall if small else Advoc)(Modes.TRAIN)
    m.subseq_len, m.train_batch_size, m.use_batchnorm = T, B, bn
    m.build(batch_size=B, seed=4)
    st = m._built
    assert st['side_on'] == side
    if not names:
      for k, lay in st['g_layers'].items():
        names[k] = [lay.kernel_name(d) for d in range(3)]
      for i, lay in enumerate(st['d_layers_fake']):
        names['layer_%d' % (i + 1)] = [lay.kernel_name(d) for d in range(3)]
    m((x, target))
    exact, noisy = {}, {}
    start = {k: st[k].clone() for k in ('g_param', 'd_param')}

    def reset():
      # every D step and G step starts from the SAME parameters (Adam would carry the atomics-order noise of a step's
      # gradients into them and no two runs would be bit-comparable any more); what the second round keeps from the first
      # is the operand images' magnitude history: one-pass images, refit checks, producer-written images
      for k, v in start.items():
        st[k].copy_(v)
      for k in ('g_m', 'g_v', 'd_m', 'd_v'):
        st[k].zero_()
      st['g_t'] = st['d_t'] = 0
      m.parameters_changed()
    for step in range(2):
      reset()
      m.d_step((x, target))
      torch.cuda.synchronize()
      for i in range(5):
        exact['%d:d:g_d_act%d' % (step, i)] = st['g_d_act'][i].clone()
      for k, v in st['d_G'].items():
        noisy['%d:%s' % (step, k)] = v.detach().clone()
      reset()
      m.g_step((x, target))
      torch.cuda.synchronize()
      for i in range(5):
        exact['%d:g:g_d_act%d' % (step, i)] = st['g_d_act'][i][B:].clone()
      for i, t in enumerate(st['g_enc']):
        exact['%d:g_enc%d' % (step, i)] = t.clone()
      for i, t in st['g_dec'].items():
        exact['%d:g_dec%d' % (step, i)] = t.clone()
      for k, v in st['g_G'].items():
        noisy['%d:%s' % (step, k)] = v.detach().clone()
    return exact, noisy

  try:
    e0, n0 = run(False)
    # Backward-data outputs that the one-stream schedule itself reproduces bit for bit (three runs): all of them except
    # where a small launch splits K over workgroups that meet in the destination with fp32 atomics (the r1 kernels of
    # AdVoc-small's 32-channel layers, a few deep layers); those few are compared like the gradients, within the noise
    reruns = [run(False)[0] for _ in range(2)]
    loose = [k for k in e0 if not all(torch.equal(e0[k], r[k]) for r in reruns)]
    assert len(loose) <= len(e0) // 3, loose
    for k in loose:
      n0[k] = e0.pop(k)
    if not small:
      assert any('patch_gemm_h3_kernel' in n for v in names.values() for n in v), names
      assert any('gather_gemm_h3_kernel' in n for v in names.values() for n in v), names
      assert any('wgrad_h3' in n for v in names.values() for n in v), names
    for trial in range(20):
      e1, n1 = run(True)
      for k in e0:
        if not torch.equal(e1[k], e0[k]):
          d = (e1[k].double() - e0[k].double()).abs()
          raise AssertionError('trial %d, bn %s: %s differs from the one-stream step in %d elements, max |d| %.3g (tensor max %.3g)'
                               % (trial, bn, k, int((d != 0).sum()), float(d.max()), float(e0[k].abs().max())))
      n1.update({k: e1[k] for k in loose})
      for k in n0:
        top = float(n0[k].abs().max())
        d = float((n1[k].double() - n0[k].double()).abs().max())
        # (a bias in front of a batch norm has an exactly-zero gradient: both sides are pure round-off of sums of ~2 M terms --
        # seen up to 1.4e-7 on either side, differences up to 1.1e-7; what the race did was 1e-3 of gradients of 1e-3 .. 1)
        assert d <= 2e-5 * top + 3e-7, (trial, bn, k, d, top)
  finally:
    monkeypatch.delenv('ADVOC_WGRAD_H3_ORDERED')
    _lib.reload_env()


@gpu
def test_weight_magnitudes_follow_the_parameters(hip, monkeypatch):
  """One advoc_segmented_amax_f32 launch per arena at the start of every forward pass replaces the per-image magnitude
  passes over the kernels (ADVOC_WEIGHT_AMAX=0 restores them): the table holds max |kernel| of the CURRENT parameters
  after optimizer steps and after an in-place edit, and the generator output does not change."""
  from advoc_amd.model import AdvocSmall, Modes
  dev = torch.device('cuda')
  x, target = batch(16, 128, 9)
  x, target = x.to(dev), target.to(dev)
  for on in (0, 1):
    monkeypatch.setenv('ADVOC_WEIGHT_AMAX', str(on))
    m = AdvocSmall(Modes.TRAIN)
    m.subseq_len = 128
    m.train_batch_size = 16
    m.build(batch_size=16, seed=4)
    st = m._built
    assert st['wamax_on'] == bool(on)
    assert all(bool(lay.struct.w_amax) == bool(on) for lay in st['g_layers'].values())
    m((x, target))
    for _ in range(3):
      m.train_loop()
    assert all(np.isfinite(v) for v in m.losses().values())
  # the last forward pass over each network ran BEFORE its last Adam step: a fresh pass brings the table up to date
  for rep in range(2):
    calls = m._dropout_calls
    out1 = m._gen_forward(st['x_in']).clone()
    m._disc_forward(st['d_layers_fake'], st['d_bn_fake'])
    for net in ('g', 'd'):
      got = st[net + '_wamax'].view(torch.float32).cpu()
      for k, i in st[net + '_wamax_index'].items():
        assert float(got[i]) == float(st[net + '_P'][k].abs().max()), (rep, k)
    saved = [(lay, lay.struct.w_amax) for lay in st['g_layers'].values()]
    assert st['g_wimg'] and st['d_wimg'] and any(lay.struct.w_img[0] for lay, _ in saved)
    for lay, _ in saved:
      lay.struct.w_amax = None                       # the layers take the magnitude and build their weight images
      for d in (0, 1):                               # themselves again
        lay.set_weight_image(d, None, None)
    m._dropout_calls = calls                         # same dropout masks
    out0 = m._gen_forward(st['x_in']).clone()
    for lay, p in saved:
      lay.struct.w_amax = p
    for lay, d, off, idx in st['g_wimg']['uses']:
      lay.set_weight_image(d, st['g_wimg']['pool'].data_ptr() + off, st['g_wimg']['hdrs'].data_ptr() + 128 * idx, l1=True)
    # same power of two, same weight images (bit-identical per layer: test_hip_conv.py); what is left is the order of the
    # split-K atomics of the small deep layers
    assert rel(out1, out0) < 2e-6, (rep, rel(out1, out0))
    for k in st['g_wamax_index']:
      st['g_P'][k].mul_(300.0 if 'encoder_2' in k else 0.01)      # an in-place edit of the views ...
    m.parameters_changed('g')                                      # ... announced: the next pass sees it


@gpu
def test_training_on_split_bf16_path_tracks_fp32_path(hip, hipenv):
  """At the benchmark geometry (32 clips x 256 frames) most contractions run on the split-bf16 matrix path
  (igemm.hip / wgrad.hip); with ADVOC_IGEMM_X6=0 ADVOC_WGRAD_X6=0 the same model runs on the fp32 MFMA
  kernels.  The first step's gradients agree to round-off, and 12 train_loops on one batch end at the same
  losses (Adam amplifies round-off, so the trajectories are compared, not the weights)."""
  from advoc_amd.model import AdvocSmall, Modes
  dev = torch.device('cuda')
  x, target = batch(32, 256, 5)
  x, target = x.to(dev), target.to(dev)

  def run(split):
    hipenv(ADVOC_IGEMM_X6=None if split else 0, ADVOC_WGRAD_X6=None if split else 0)
    m = AdvocSmall(Modes.TRAIN)
    m.train_batch_size = 32
    m.build(batch_size=32, seed=11)
    m((x, target))
    m.train_loop()
    torch.cuda.synchronize()
    st = m._built
    grads = {k: v.detach().clone() for name in ('d_G', 'g_G') for k, v in st[name].items()}
    names = set(l.kernel_name(0) for l in st['g_layers'].values())
    hist = [m.losses()]
    for _ in range(11):
      m.train_loop()
      hist.append(m.losses())
    return grads, hist, names

  g_s, h_s, n_s = run(True)
  g_f, h_f, n_f = run(False)
  split = lambda n: (n.startswith('gather_gemm_kernel<') and n.endswith(', true>')) or 'h3' in n       # noqa: E731
  assert any(split(n) for n in n_s) and not any(split(n) for n in n_f), (n_s, n_f)
  worst = 0.0
  for k in g_f:
    if not k.startswith('discriminator'):      # D gradients are taken at identical weights
      continue
    e = rel(g_s[k], g_f[k])
    worst = max(worst, e)
    assert e < 1e-4, (k, e)          # measured 1.5e-5 (the deepest D layers; single convs agree to 3e-6)
  for a, b in zip(h_s, h_f):
    assert all(np.isfinite(v) for v in a.values()), a
  for key in ('gen_loss_L1', 'disc_loss'):
    a, b = h_s[-1][key], h_f[-1][key]
    assert abs(a - b) <= 0.02 * abs(b) + 1e-3, (key, a, b)
  print('split vs fp32: first-step D gradients %.2e; final L1 %.5f vs %.5f, D loss %.5f vs %.5f' % (
      worst, h_s[-1]['gen_loss_L1'], h_f[-1]['gen_loss_L1'], h_s[-1]['disc_loss'], h_f[-1]['disc_loss']))


@gpu
def test_batch_norm_split_entry_points_reproduce_global_statistics(hip):
  """advoc_bn_*_stats on two shards + a sum of the 2c doubles + *_finalize / *_apply with the global
  count == the one-call batch norm on the whole batch (what synchronised BN across replicas relies on)."""
  from advoc_amd import _lib
  lib = _lib.load()
  dev = torch.device('cuda')
  g_ = torch.Generator().manual_seed(2)
  n, c = 6 * 5 * 7, 64
  z = (torch.randn(n, c, generator=g_) * 2 + 3).to(dev)
  grad = torch.randn(n, c, generator=g_).to(dev)
  gamma = (1 + 0.1 * torch.randn(c, generator=g_)).to(dev)
  beta = (0.1 * torch.randn(c, generator=g_)).to(dev)
  new = lambda: [torch.zeros(c, device=dev) for _ in range(4)]            # noqa: E731  scale shift mean invstd
  # reference: one call on everything
  sc, sh, mu, isd = new()
  work = torch.zeros(4 * c, device=dev)
  _lib.check(lib.advoc_bn_forward(_lib.ptr(z), n, c, _lib.ptr(gamma), _lib.ptr(beta), 1e-5, _lib.ptr(sc), _lib.ptr(sh),
                                  _lib.ptr(mu), _lib.ptr(isd), _lib.ptr(work), _lib.stream()), 'fwd')
  g_all = grad.clone()
  dga, dbe = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
  _lib.check(lib.advoc_bn_backward(_lib.ptr(z), _lib.ptr(g_all), n, c, _lib.ptr(gamma), _lib.ptr(mu), _lib.ptr(isd),
                                   _lib.ptr(dga), _lib.ptr(dbe), 0, _lib.ptr(work), _lib.stream()), 'bwd')
  # two shards of unequal size
  cut = 80
  shards = [(z[:cut].contiguous(), grad[:cut].clone()), (z[cut:].contiguous(), grad[cut:].clone())]
  works = [torch.zeros(4 * c, device=dev) for _ in shards]
  for (zs, _), w in zip(shards, works):
    _lib.check(lib.advoc_bn_forward_stats(_lib.ptr(zs), zs.shape[0], c, _lib.ptr(w), _lib.stream()), 'stats')
  tot = works[0].view(torch.float64) + works[1].view(torch.float64)
  sc2, sh2, mu2, isd2 = new()
  wsum = tot.view(torch.float32).contiguous()
  _lib.check(lib.advoc_bn_forward_finalize(_lib.ptr(wsum), n, c, _lib.ptr(gamma), _lib.ptr(beta), 1e-5, _lib.ptr(sc2),
                                           _lib.ptr(sh2), _lib.ptr(mu2), _lib.ptr(isd2), _lib.stream()), 'finalize')
  for a, b in ((sc, sc2), (sh, sh2), (mu, mu2), (isd, isd2)):
    assert rel(b, a) < 1e-6
  dga2, dbe2 = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
  for i, ((zs, gs), w) in enumerate(zip(shards, works)):
    _lib.check(lib.advoc_bn_backward_stats(_lib.ptr(zs), _lib.ptr(gs), zs.shape[0], c, _lib.ptr(mu2), _lib.ptr(isd2),
                                           _lib.ptr(dga2), _lib.ptr(dbe2), int(i > 0), _lib.ptr(w), _lib.stream()), 'bstats')
  tot = (works[0].view(torch.float64) + works[1].view(torch.float64)).view(torch.float32).contiguous()
  for zs, gs in shards:
    _lib.check(lib.advoc_bn_backward_apply(_lib.ptr(zs), _lib.ptr(gs), zs.shape[0], c, _lib.ptr(gamma), _lib.ptr(mu2),
                                           _lib.ptr(isd2), _lib.ptr(tot), n, _lib.stream()), 'apply')
  assert rel(torch.cat([shards[0][1], shards[1][1]]), g_all) < 1e-5
  assert close(dga2, dga, 1e-5) and close(dbe2, dbe, 1e-5)


@gpu
@pytest.mark.parametrize('bn', [False, True])
def test_width_override_ngf96_ndf64(hip, bn):
  """--model_overrides "ngf=96,ndf=64": channel counts that are multiples of 32 but not powers of two
  (96, 192, 384, 768) through every kernel family, one full G step against the float64 oracle."""
  cfg, P, m = make(True, 32, 2, seed=2, ngf=96, ndf=64, bn=bn)
  x, target = batch(2, 32, 8)
  masks = A.make_dropout_masks(cfg, 2, seed=4)
  m.set_dropout_masks(dev_masks(masks))
  dev = torch.device('cuda')
  P64 = {k: v.double() for k, v in P.items()}
  m64 = {k: v.double() for k, v in masks.items()}
  gG, LG = A.grads(P64, x.double(), target.double(), cfg, m64, 'G')
  gG32, _ = A.grads(P, x, target, cfg, masks, 'G')
  m._lr = 0.0
  m.g_step((x.to(dev), target.to(dev)))
  assert abs(m.losses()['gen_loss_total'] - float(LG['g_loss'])) < 1e-4 * abs(float(LG['g_loss']))
  st = m._built
  # with batch norm at this size one leaky-ReLU gate within round-off of 0 moves a gradient by ~1e-3
  # (see test_train_loop_matches_oracle_trainer); the kernels themselves are exact at these widths
  bar = 10 * GRAD_BAR if bn else GRAD_BAR
  for k, v in gG.items():
    assert close(st['g_G'][k], v, max(bar, 3 * rel(gG32[k], v))), (k, rel(st['g_G'][k], v), rel(gG32[k], v))


@gpu
def test_losses_saturate_like_the_tf_formulas(hip):
  """Extreme logits: sigmoid saturates to exactly 0 / 1 in fp32 and the EPS = 1e-12 inside the logs is
  what keeps the losses finite (advoc_model.py:8,238-241); values and gradients must follow those
  literal formulas (autograd of the same fp32 expressions), not a numerically 'nicer' softplus."""
  from advoc_amd import _lib
  lib = _lib.load()
  dev = torch.device('cuda')
  zr = torch.tensor([-100.0, -20.0, -1.5, 0.0, 0.7, 20.0, 100.0, 3.0, -3.0], requires_grad=True)
  zf = torch.tensor([100.0, 20.0, 1.5, 0.0, -0.7, -20.0, -100.0, -3.0, 3.0], requires_grad=True)
  n = zr.numel()
  p_r, p_f = torch.sigmoid(zr), torch.sigmoid(zf)
  d_loss = torch.mean(-(torch.log(p_r + 1e-12) + torch.log(1 - p_f + 1e-12)))
  gr, gf = torch.autograd.grad(d_loss, [zr, zf])
  dzr, dzf, sums = torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(4, device=dev)
  zr_d, zf_d = zr.detach().to(dev), zf.detach().to(dev)       # named: _lib.ptr() does not keep its tensor alive
  _lib.check(lib.advoc_gan_d_loss(_lib.ptr(zr_d), _lib.ptr(zf_d), n, _lib.ptr(dzr),
                                  _lib.ptr(dzf), _lib.ptr(sums[0:1]), _lib.stream()), 'd_loss')
  assert np.isfinite(float(sums[0])) and abs(float(sums[0]) / n - float(d_loss.detach())) < 1e-5 * abs(float(d_loss.detach()))
  assert torch.allclose(dzr.cpu(), gr, rtol=1e-5, atol=1e-12) and torch.allclose(dzf.cpu(), gf, rtol=1e-5, atol=1e-12)
  # generator side
  zf2 = zf.detach().clone().requires_grad_(True)
  g_gan = torch.mean(-torch.log(torch.sigmoid(zf2) + 1e-12))
  (gg,) = torch.autograd.grad(g_gan, [zf2])
  gen, tgt = torch.rand(1, 4, 513, 1), torch.rand(1, 4, 513, 1)
  dz, dgen, s2 = torch.zeros(n, device=dev), torch.zeros(gen.shape, device=dev), torch.zeros(4, device=dev)
  gen_d, tgt_d = gen.to(dev), tgt.to(dev)
  _lib.check(lib.advoc_gan_g_loss(_lib.ptr(zf_d), n, _lib.ptr(gen_d), _lib.ptr(tgt_d),
                                  gen.numel(), 1.0, 10.0, _lib.ptr(dz), _lib.ptr(dgen), 0, _lib.ptr(s2[0:2]),
                                  _lib.stream()), 'g_loss')
  assert abs(float(s2[0]) / n - float(g_gan.detach())) < 1e-5 * abs(float(g_gan.detach()))
  assert torch.allclose(dz.cpu(), gg, rtol=1e-5, atol=1e-12)
  assert abs(float(s2[1]) / gen.numel() - float((tgt - gen).abs().mean())) < 1e-6
  want_dgen = -10.0 / gen.numel() * torch.sign(tgt - gen)
  assert torch.allclose(dgen.cpu(), want_dgen, rtol=1e-6, atol=0)
