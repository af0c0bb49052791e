"""Every conv / transposed-conv direction of the C ABI against the torch-CPU oracle
(oracle/advoc_torch.py), layer shapes mirroring each layer type of AdVoc / AdVoc-small at
reduced size, including the odd widths (SAME pad (1,2)), the skip concat, the [:, :, :-1, :]
trims, dropout masks and the thin edge layers.  Tolerance: fp32 summation-order noise."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import advoc_torch as A

gpu = pytest.mark.gpu
TOL = 2e-5


def rel(a, b):
  a = a.detach().double().cpu()
  b = b.detach().double().cpu()
  return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _act(x, act):
  if act == 1:
    return A.lrelu(x)
  if act == 2:
    return torch.relu(x)
  return x


def oracle_layer(kind, x0, x1, in_w, w, b, stride, pad, act, mask, keep, out_w, dy):
  """Forward + all gradients in float64 on the CPU.  Returns y, dx0, dx1, dw, db."""
  x0 = x0.double().requires_grad_(True)
  x1 = x1.double().requires_grad_(True) if x1 is not None else None
  w = w.double().requires_grad_(True)
  b = b.double().requires_grad_(True)
  parts = [x0[:, :, :in_w, :]] + ([x1[:, :, :in_w, :]] if x1 is not None else [])
  inp = _act(torch.cat(parts, dim=3), act)
  if kind == 0:
    H, W = inp.shape[1], inp.shape[2]
    # explicit pads: top/left as given, bottom/right whatever the output size needs
    oh, ow = dy.shape[1], out_w
    pb = max((oh - 1) * stride[0] + w.shape[0] - H - pad[0], 0)
    pr = max((ow - 1) * stride[1] + w.shape[1] - W - pad[1], 0)
    xp = torch.nn.functional.pad(inp.permute(0, 3, 1, 2), (pad[1], pr, pad[0], pb))
    y = torch.nn.functional.conv2d(xp, w.permute(3, 2, 0, 1).contiguous(), b, stride=stride).permute(0, 2, 3, 1)
    y = y[:, :oh, :ow]
  else:
    y = A.gen_deconv(inp, w, b, strides=stride)[:, :, :out_w, :]
  if mask is not None:
    y = (y / keep) * mask.double()[:, :, :out_w]
  g = torch.autograd.grad(y, [x0, w, b] + ([x1] if x1 is not None else []), dy.double()[:, :, :out_w])
  return y.detach(), g[0], (g[3] if x1 is not None else None), g[1], g[2]


CASES = [
    # name, kind, (B,H,W), c0, c1, cout, trim, stride, pad(None=SAME), act, dropout, clip_out
    ('enc_same_odd',   0, (2, 16, 33), 32, 0, 64, 0, (2, 2), None, 1, False, 0),
    ('enc_same_wide',  0, (2, 8, 17), 128, 0, 128, 0, (2, 2), None, 1, False, 0),
    ('enc_n256',       0, (1, 8, 9), 64, 0, 256, 0, (2, 2), None, 1, False, 0),
    ('enc_n32',        0, (3, 10, 13), 32, 0, 32, 0, (2, 2), None, 1, False, 0),
    ('enc1_cin1',      0, (2, 32, 65), 1, 0, 32, 0, (2, 2), None, 0, False, 0),
    ('dec_first_drop', 1, (2, 4, 9), 64, 0, 64, 0, (2, 2), (1, 1), 2, True, 0),
    ('dec_skip_trim',  1, (2, 8, 17), 64, 64, 32, 1, (2, 2), (1, 1), 2, False, 0),
    ('dec_skip_drop',  1, (2, 8, 17), 128, 64, 128, 1, (2, 2), (1, 1), 2, True, 0),
    ('dec1_cout1',     1, (2, 16, 33), 32, 32, 1, 1, (2, 2), (1, 1), 2, False, 1),
    ('dec1_cout1_big', 1, (1, 8, 9), 128, 128, 1, 1, (2, 2), (1, 1), 2, False, 1),
    ('d1_cin2',        0, (2, 32, 65), 1, 1, 32, 0, (2, 2), (1, 1), 0, False, 0),
    ('d2_even',        0, (2, 16, 32), 32, 0, 64, 0, (2, 2), (1, 1), 1, False, 0),
    ('d4_s1',          0, (2, 9, 12), 64, 0, 128, 0, (1, 1), (1, 1), 1, False, 0),
    ('d5_cout1',       0, (2, 9, 11), 128, 0, 1, 0, (1, 1), (1, 1), 1, False, 0),
    # the fused one-launch form of the <= 2-column layers (fused_taps_kernel): many patches, partial patches, two columns
    ('dec1_cout1_tiles', 1, (2, 40, 70), 32, 32, 1, 1, (2, 2), (1, 1), 2, False, 1),
    ('d5_cout1_tiles', 0, (2, 40, 45), 64, 0, 1, 0, (1, 1), (1, 1), 1, False, 0),
    ('d1_cin2_k64',    0, (2, 32, 65), 1, 1, 64, 0, (2, 2), (1, 1), 0, False, 0),
    ('d1_cin2_k128_tiles', 0, (2, 64, 129), 1, 1, 128, 0, (2, 2), (1, 1), 0, False, 0),
    # time axis collapsed (subseq_len 64 with 8 encoders, advoc_model.py:109-116,139-142): strides (1,2)
    ('enc_s12_h1',     0, (3, 1, 9), 64, 0, 64, 0, (1, 2), None, 1, False, 0),
    ('enc_s12_h2',     0, (2, 2, 17), 32, 0, 64, 0, (1, 2), None, 1, False, 0),
    ('dec_s12_first',  1, (3, 1, 5), 64, 0, 64, 0, (1, 2), (1, 1), 2, True, 0),
    ('dec_s12_skip',   1, (2, 1, 9), 64, 64, 32, 1, (1, 2), (1, 1), 2, False, 0),
]


def build_case(case, seed=0):
  name, kind, (B, H, W), c0, c1, cout, trim, stride, pad, act, drop, clip = case
  g = torch.Generator().manual_seed(seed)
  x0 = torch.randn(B, H, W + trim, c0, generator=g)
  x1 = torch.randn(B, H, W, c1, generator=g) if c1 else None
  cin = c0 + c1
  if kind == 0:
    if pad is None:
      pt, _ = A.same_pad(H, 4, stride[0])
      pl, _ = A.same_pad(W, 4, stride[1])
      oh, ow = -(-H // stride[0]), -(-W // stride[1])
    else:
      pt, pl = pad
      oh = (H + 2 * pt - 4) // stride[0] + 1
      ow = (W + 2 * pl - 4) // stride[1] + 1
    w = torch.randn(4, 4, cin, cout, generator=g) * 0.05
  else:
    pt, pl = 1, 1
    oh, ow = stride[0] * H, stride[1] * W - clip
    w = torch.randn(4, 4, cout, cin, generator=g) * 0.05
  b = torch.randn(cout, generator=g) * 0.1
  mask = (torch.rand(B, oh, ow, cout, generator=g) >= 0.5).to(torch.uint8) if drop else None
  dy = torch.randn(B, oh, ow, cout, generator=g)
  return dict(kind=kind, x0=x0, x1=x1, in_w=W, w=w, b=b, stride=stride, pad=(pt, pl), act=act,
              mask=mask, keep=0.5, out_w=ow, dy=dy, oh=oh)


THIN = [c for c in CASES if c[0] in ('dec1_cout1', 'dec1_cout1_big', 'd1_cin2', 'd5_cout1')]


@gpu
@pytest.mark.parametrize('case', THIN, ids=[c[0] for c in THIN])
def test_layer_all_directions_without_workspace(hip, case):
  """The 1-2 channel layers have a direct kernel for callers that pass no scratch buffer."""
  test_layer_all_directions(hip, case, workspace=False)


SPLITK = [c for c in CASES if c[0] in ('enc_same_wide', 'enc_n256', 'dec_first_drop', 'dec_skip_drop', 'd4_s1')]


@gpu
@pytest.mark.parametrize('case', SPLITK, ids=[c[0] for c in SPLITK])
def test_layer_single_k_pass(hip, case, hipenv):
  """Small pixel grids split the contraction over workgroups (atomics) by default; with
  ADVOC_IGEMM_SPLITK=0 the same layers run one K pass.  Both must meet the bar, and the single-pass
  result must be bitwise reproducible."""
  from advoc_amd import conv
  hipenv(ADVOC_IGEMM_SPLITK=0)
  test_layer_all_directions(hip, case)
  c = build_case(case)
  dev = torch.device('cuda')
  x0 = c['x0'].to(dev)
  x1 = c['x1'].to(dev) if c['x1'] is not None else None
  w = c['w'].to(dev)
  cout = w.shape[3] if c['kind'] == 0 else w.shape[2]
  ys = []
  for _ in range(2):
    y = torch.zeros(x0.shape[0], c['oh'], c['out_w'], cout, device=dev)
    conv.Layer(c['kind'], x0, y, w, None, x1=x1, in_w=c['in_w'], stride=c['stride'], pad=c['pad'],
               in_act=c['act']).forward()
    ys.append(y)
  assert torch.equal(ys[0], ys[1])


# Launches of >= 512 tiles whose count is not a multiple of the CU count: the last tiles are cut
# into K slices, parked in the workspace and summed by the last slice to arrive (igemm.hip).
# enc_tail: 537 forward tiles, 1076 backward-data tiles (4 phases); dec_tail_drop: 1092 / 546.
TAIL = [
    ('enc_tail',      0, (65, 32, 65), 32, 0, 64, 0, (2, 2), None, 1, False, 0),
    ('dec_tail_drop', 1, (33, 16, 33), 64, 64, 64, 1, (2, 2), (1, 1), 2, True, 0),
]


@gpu
@pytest.mark.parametrize('case', TAIL, ids=[c[0] for c in TAIL])
def test_layer_tail_split(hip, case, hipenv):
  import ctypes
  from advoc_amd import _lib, conv
  c = build_case(case)
  dev = torch.device('cuda')
  x0 = c['x0'].to(dev)
  x1 = c['x1'].to(dev) if c['x1'] is not None else None
  w = c['w'].to(dev)
  cout = w.shape[3] if c['kind'] == 0 else w.shape[2]

  def forward():
    y = torch.full((x0.shape[0], c['oh'], c['out_w'], cout), float('nan'), device=dev)
    L = conv.Layer(c['kind'], x0, y, w, None, x1=x1, in_w=c['in_w'], stride=c['stride'], pad=c['pad'],
                   in_act=c['act'])
    L.forward()
    return L, y

  L, y_a = forward()
  # the path under test is really taken: both directions ask for scratch, and get it
  for direction in (0, 1):
    assert _lib.load().advoc_conv_workspace_bytes(ctypes.byref(L.struct), direction) > 0
  assert L.struct.workspace_bytes > 0
  _, y_b = forward()
  assert torch.equal(y_a, y_b)                      # fixed summation order: run-to-run reproducible
  hipenv(ADVOC_IGEMM_TAIL=0)
  _, y_plain = forward()
  assert rel(y_a, y_plain.double()) < 1e-6          # same numbers up to the order of one sum
  hipenv(ADVOC_IGEMM_TAIL=None)
  test_layer_all_directions(hip, case)              # all three directions against the float64 oracle
  test_layer_all_directions(hip, case, workspace=False)


# Launches of >= 448 tiles of 128 rows run on the bf16 matrix cores with every fp32 operand split exactly
# into three bf16 terms (igemm.hip, wgrad.hip): same tolerance as the fp32 MFMA kernels, and the two paths
# agree with each other far below it.
SPLIT = [
    ('enc_split',      0, (32, 64, 129), 64, 0, 128, 0, (2, 2), None, 1, False, 0),
    # 455 tiles: the split kernels together with split-K (atomics) in the forward pass
    ('enc_split_k',    0, (28, 64, 129), 64, 0, 128, 0, (2, 2), None, 1, False, 0),
    ('dec_split_drop', 1, (16, 32, 65), 64, 64, 128, 1, (2, 2), (1, 1), 2, True, 0),
]


@gpu
@pytest.mark.parametrize('case', SPLIT, ids=[c[0] for c in SPLIT])
def test_layer_split_bf16_path(hip, case, hipenv):
  from advoc_amd import conv
  c = build_case(case)
  dev = torch.device('cuda')
  x0 = c['x0'].to(dev)
  x1 = c['x1'].to(dev) if c['x1'] is not None else None
  w, dy = c['w'].to(dev), c['dy'].to(dev)
  cout = w.shape[3] if c['kind'] == 0 else w.shape[2]

  def run():
    y = torch.full((x0.shape[0], c['oh'], c['out_w'], cout), float('nan'), device=dev)
    L = conv.Layer(c['kind'], x0, y, w, None, x1=x1, in_w=c['in_w'], stride=c['stride'], pad=c['pad'],
                   in_act=c['act'])
    L.forward()
    dx0 = torch.full_like(x0, 7.0)
    dx1 = torch.full_like(x1, 7.0) if x1 is not None else None
    L.backward_data(dy, dx0, dx1)
    dw = torch.full_like(w, float('nan'))
    L.backward_weight(dy, dw)
    return [L.kernel_name(d) for d in range(3)], y, dx0[:, :, :c['in_w']].clone(), dw

  def split(n):
    return n.endswith(', true>') or 'h3' in n
  hipenv(ADVOC_H3=0)                                                # register-split kernels (igemm.hip, wgrad.hip)
  names, y, dx, dw = run()
  assert all(split(n) and 'h3' not in n for n in names), names     # the split kernels are the ones that ran
  hipenv(ADVOC_IGEMM_X6=0, ADVOC_WGRAD_X6=0)
  names32, y32, dx32, dw32 = run()
  assert not any(split(n) for n in names32), names32
  for a, b in ((y, y32), (dx, dx32), (dw, dw32)):
    assert rel(a, b) < 3e-6, rel(a, b)
  hipenv(ADVOC_IGEMM_X6=None, ADVOC_WGRAD_X6=None)
  test_layer_all_directions(hip, case)              # all three directions against the float64 oracle
  # operand-image kernels (igemm_h3.hip: fp16 pairs, three products): agree with the register-split result far
  # below the oracle tolerance, and meet the oracle themselves
  hipenv(ADVOC_H3=1, ADVOC_H3_MIN_TILES=1)
  names_d, y_d, dx_d, _ = run()
  assert 'h3' in names_d[0], names_d      # (backward-data has 64 columns here: register-split kernel)
  assert rel(y_d, y) < 3e-6 and rel(dx_d, dx) < 3e-6, (rel(y_d, y), rel(dx_d, dx))
  hipenv(ADVOC_H3=None, ADVOC_H3_MIN_TILES=None)
  test_layer_all_directions(hip, case)


# Operand-image kernels (igemm_h3.hip): every tile / stage instance against the float64 oracle, in all
# directions that take them (forward, backward-data), on shapes with the features the loader has to get right:
# rows that are not a multiple of the tile, two sources with a trimmed column, SAME padding on odd widths,
# sub-pixel phases, stride-1 taps, dropout on either side.  ADVOC_H3_MIN_TILES=1 lets these small launches in.
H3 = [
    ('h3_enc',       0, (3, 16, 33), 128, 0, 256, 0, (2, 2), None, 1, False, 0),
    ('h3_enc_s1',    0, (2, 9, 12), 128, 0, 128, 0, (1, 1), (1, 1), 1, False, 0),
    ('h3_dec_skip',  1, (2, 8, 17), 128, 128, 256, 1, (2, 2), (1, 1), 2, True, 0),
    ('h3_dec_first', 1, (3, 4, 9), 256, 0, 128, 0, (2, 2), (1, 1), 2, True, 0),
    ('h3_enc_s12',   0, (3, 2, 17), 128, 0, 128, 0, (1, 2), None, 1, False, 0),
    ('h3_dec_mixed', 1, (2, 8, 9), 256, 128, 128, 1, (2, 2), (1, 1), 2, False, 0),
    # 64 / 192 columns: the 128 x 64 tile whatever is asked for; 32-channel sources
    ('h3_enc_n64',    0, (3, 16, 33), 32, 0, 64, 0, (2, 2), None, 1, False, 0),
    ('h3_dec_n192',   1, (2, 8, 17), 64, 32, 192, 1, (2, 2), (1, 1), 2, True, 0),
    # deep contraction on a small grid: split-K slices meeting with atomics
    ('h3_deep_small', 0, (2, 8, 9), 512, 0, 128, 0, (2, 2), None, 1, False, 0),
]
# 32 output columns (AdVoc-small: encoder_2 / layer_2 backward-data -- 32 input channels --, decoder_2 forward -- 32 output
# channels; advoc_model_small.py:14-15): the 128 x 64 tile with its upper 32 columns masked (r4; the r1 fp32 kernel before)
H3_N32 = [
    (('h3_small_enc2', 0, (3, 32, 33), 32, 0, 64, 0, (2, 2), None, 1, False, 0), {0: 'gather_gemm_h3_kernel<2, 1, 3, 2>', 1: 'gather_gemm_h3_kernel<2, 1, 3, 2>'}),
    (('h3_small_dec2', 1, (2, 16, 17), 64, 64, 32, 1, (2, 2), (1, 1), 2, False, 0), {0: 'gather_gemm_h3_kernel<2, 1, 3, 2>'}),
    (('h3_small_dec2_drop', 1, (2, 8, 9), 64, 64, 32, 1, (2, 2), (1, 1), 2, True, 1), {0: 'gather_gemm_h3_kernel<2, 1, 3, 2>'}),
    (('h3_small_layer2', 0, (4, 16, 32), 32, 0, 64, 0, (2, 2), (1, 1), 1, False, 0), {1: 'gather_gemm_h3_kernel<2, 1, 3, 2>'}),
]
H3_VARIANTS = [(1, 2), (4, 3), (4, 2), (5, 2)]   # 128 x 128, 128 x 64 (three stages: the default since r6; two), 256 x 256 on 8 waves

# Patch kernels (igemm_patch.hip): the stride-1 gathers -- four fused sub-pixel phases (transposed-conv forward, conv
# backward-data) and the 4x4 stride-1 conv in both directions -- on grids that are not multiples of the 16 x 16 patch,
# with two sources and a trimmed column, dropout on either side, several column tiles.  (direction, kernel) expected:
PATCH = [
    (('p3_dec_skip',  1, (2, 31, 30), 64, 64, 128, 1, (2, 2), (1, 1), 2, True, 0), {0: 'patch_gemm_h3_kernel<4, 0>'}),
    (('p3_dec_clip',  1, (3, 30, 31), 96, 0, 64, 0, (2, 2), (1, 1), 2, False, 1), {0: 'patch_gemm_h3_kernel<4, 0>'}),
    (('p3_enc_bwd',   0, (2, 62, 60), 64, 0, 128, 0, (2, 2), None, 1, False, 0), {1: 'patch_gemm_h3_kernel<4, 1>'}),
    (('p3_enc_bwd_odd', 0, (2, 61, 59), 128, 0, 64, 0, (2, 2), None, 1, True, 0), {1: 'patch_gemm_h3_kernel<4, 1>'}),
    (('p3_d4',        0, (2, 32, 31), 256, 0, 256, 0, (1, 1), (1, 1), 1, False, 0),
     {0: 'patch_gemm_h3_kernel<1, 0>', 1: 'patch_gemm_h3_kernel<1, 1>'}),
    (('p3_d4_wide',   0, (1, 29, 33), 64, 0, 512, 0, (1, 1), (1, 1), 1, True, 0), {0: 'patch_gemm_h3_kernel<1, 0>'}),
    # stride-2 gathers as four parity planes of the input (<2, .>: 256 columns per workgroup, <3, .>: 128): conv forward on
    # even and odd input sizes (SAME padding), transposed-conv backward-data into two destinations
    (('p3_s2_enc',    0, (2, 62, 60), 64, 0, 256, 0, (2, 2), None, 1, False, 0),
     {0: 'patch_gemm_h3_kernel<2, 0>', 1: 'patch_gemm_h3_kernel<4, 1>'}),
    (('p3_s2_enc_odd', 0, (2, 61, 59), 32, 0, 128, 0, (2, 2), None, 1, True, 0), {0: 'patch_gemm_h3_kernel<3, 0>'}),
    (('p3_s2_enc_pad', 0, (1, 62, 64), 96, 0, 384, 0, (2, 2), (1, 1), 0, False, 0), {0: 'patch_gemm_h3_kernel<3, 0>'}),
    (('p3_s2_dec_bwd', 1, (2, 31, 30), 128, 128, 64, 1, (2, 2), (1, 1), 2, True, 0),
     {0: 'patch_gemm_h3_kernel<4, 0>', 1: 'patch_gemm_h3_kernel<2, 1>'}),
    (('p3_s2_dec_bwd128', 1, (3, 30, 31), 128, 0, 64, 0, (2, 2), (1, 1), 2, False, 1),
     {0: 'patch_gemm_h3_kernel<4, 0>', 1: 'patch_gemm_h3_kernel<3, 1>'}),
    # grids of 16 n + 1..3 columns (the model's 33 / 65 / 129): patches over the multiple of 16, a per-tap launch over
    # the remaining columns
    (('p3_rem_dec',   1, (2, 16, 33), 64, 64, 128, 1, (2, 2), (1, 1), 2, True, 0),
     {0: 'patch_gemm_h3_kernel<4, 0>', 1: 'patch_gemm_h3_kernel<3, 1>'}),
    (('p3_rem_enc',   0, (2, 32, 65), 64, 0, 256, 0, (2, 2), None, 1, False, 0),
     {0: 'patch_gemm_h3_kernel<2, 0>', 1: 'patch_gemm_h3_kernel<4, 1>'}),
    (('p3_rem_d4',    0, (1, 33, 36), 64, 0, 256, 0, (1, 1), (1, 1), 1, True, 0), {0: 'patch_gemm_h3_kernel<1, 0>'}),
    # (r4) 32 columns on the 64-column four-phase instance (upper half of the weight tile masked in the DMA, never stored):
    # AdVoc-small's decoder_2 forward (two sources, trimmed column, clipped output) and encoder_2 / layer_2 backward-data
    (('p3_n32_dec',   1, (2, 31, 30), 64, 64, 32, 1, (2, 2), (1, 1), 2, False, 1), {0: 'patch_gemm_h3_kernel<4, 0>'}),
    (('p3_n32_dec_drop', 1, (2, 16, 33), 64, 64, 32, 1, (2, 2), (1, 1), 2, True, 0), {0: 'patch_gemm_h3_kernel<4, 0>'}),
    (('p3_n32_enc_bwd', 0, (2, 62, 60), 32, 0, 64, 0, (2, 2), None, 1, False, 0), {1: 'patch_gemm_h3_kernel<4, 1>'}),
    (('p3_n32_d2_bwd', 0, (3, 64, 66), 32, 0, 64, 0, (2, 2), (1, 1), 1, True, 0), {1: 'patch_gemm_h3_kernel<4, 1>'}),
    # (r4) the 4x4 stride-1 gather with 128 columns per workgroup (<5, .>): AdVoc-small's layer_4 backward-data (256 -> 128
    # channels) and a 128-channel forward; ragged grids, dropout, a 16 n + 3 wide grid with its remainder columns
    (('p3_d4_small',  0, (2, 32, 31), 128, 0, 256, 0, (1, 1), (1, 1), 1, False, 0),
     {0: 'patch_gemm_h3_kernel<1, 0>', 1: 'patch_gemm_h3_kernel<5, 1>'}),
    (('p3_d4_n128',   0, (1, 29, 33), 64, 0, 128, 0, (1, 1), (1, 1), 1, True, 0), {0: 'patch_gemm_h3_kernel<5, 0>'}),
    (('p3_rem_d4_n384', 0, (1, 33, 36), 128, 0, 384, 0, (1, 1), (1, 1), 1, False, 0), {0: 'patch_gemm_h3_kernel<5, 0>'}),
]


def two_per_cu_name(name, n_cols, twowg, direction):
  """(r6) the four-phase gathers with whole 64-column tiles run as patch_gemm_h3_kernel<6, .> -- two 4-wave workgroups per
  CU -- under ADVOC_H3_PATCH_2WG: 1 the backward-data launches, 2 the forward ones too, 0 (default) none."""
  if name.startswith('patch_gemm_h3_kernel<4,') and n_cols % 64 == 0 and (twowg == 2 or (twowg == 1 and direction == 1)):
    return name.replace('<4,', '<6,')
  return name


@gpu
@pytest.mark.parametrize('twowg', [0, 1, 2], ids=['one_per_cu', 'two_per_cu_bwd', 'two_per_cu_all'])
@pytest.mark.parametrize('persist', [2, 1, 0], ids=['persistent', 'persistent_forward_only', 'tile_per_wg'])
@pytest.mark.parametrize('case,want', PATCH, ids=[c[0][0] for c in PATCH])
def test_layer_patch_kernels(hip, case, want, persist, twowg, hipenv):
  from advoc_amd import conv
  if twowg != 0 and not any(n.startswith('patch_gemm_h3_kernel<4,') for n in want.values()):
    pytest.skip('no four-phase launch in this case: the default form covers it')
  hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_H3_PATCH_MIN_WGS=1, ADVOC_H3_PATCH_PERSIST=persist, ADVOC_H3_PATCH_2WG=twowg)
  c = build_case(case)
  dev = torch.device('cuda')
  x0 = c['x0'].to(dev)
  x1 = c['x1'].to(dev) if c['x1'] is not None else None
  w = c['w'].to(dev)
  cout = w.shape[3] if c['kind'] == 0 else w.shape[2]
  cin_all = x0.shape[3] + (x1.shape[3] if x1 is not None else 0)
  y = torch.empty(x0.shape[0], c['oh'], c['out_w'], cout, device=dev)
  L = conv.Layer(c['kind'], x0, y, w, None, x1=x1, in_w=c['in_w'], stride=c['stride'], pad=c['pad'], in_act=c['act'])
  for direction, name in want.items():
    name = two_per_cu_name(name, cout if direction == 0 else cin_all, twowg, direction)
    assert L.kernel_name(direction) == name, (direction, L.kernel_name(direction), name)
  test_layer_all_directions(hip, case)
  # and the per-tap tiles on the same shapes agree with it far below the oracle tolerance
  outs = {}
  for patch in (1, 0):
    hipenv(ADVOC_H3_PATCH=patch)
    L = conv.Layer(c['kind'], x0, y, w, None, x1=x1, in_w=c['in_w'], stride=c['stride'], pad=c['pad'], in_act=c['act'])
    if patch == 0:
      assert not any('patch' in L.kernel_name(d) for d in (0, 1))
    L.forward()
    dx0 = torch.zeros_like(x0)
    dx1 = torch.zeros_like(x1) if x1 is not None else None
    L.backward_data(c['dy'].to(dev), dx0, dx1)
    outs[patch] = (y.clone(), dx0[:, :, :c['in_w']].clone())
  for a, b in zip(outs[1], outs[0]):
    assert rel(a, b) < 2e-6, rel(a, b)


@gpu
@pytest.mark.parametrize('mode', ['workspace', 'atomics', 'one_pass'])
@pytest.mark.parametrize('case,want', [c for c in PATCH if c[0][0].startswith('p3_rem')], ids=lambda c: c[0] if isinstance(c, tuple) else '')
def test_patch_remainder_columns_k_split(hip, case, want, mode, hipenv):
  """The per-tap launch that takes the 1..4 grid columns a patch launch leaves over: K slices meeting in the workspace
  (default), in the destination with atomics (ADVOC_H3_REM_WS=0), or no split at all -- same results."""
  if mode == 'workspace':
    hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_H3_PATCH_MIN_WGS=1, ADVOC_H3_REM_WS=1, ADVOC_H3_REM_WGS_PER_CU=4, ADVOC_H3_REM_SPLIT_DIV=2)
  elif mode == 'atomics':
    hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_H3_PATCH_MIN_WGS=1, ADVOC_H3_REM_WS=0, ADVOC_H3_REM_WGS_PER_CU=4, ADVOC_H3_REM_SPLIT_DIV=2)
  else:
    hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_H3_PATCH_MIN_WGS=1, ADVOC_IGEMM_SPLITK=0)
  test_layer_all_directions(hip, case)


@gpu
@pytest.mark.parametrize('variant', H3_VARIANTS, ids=['t%d_s%d' % v for v in H3_VARIANTS])
@pytest.mark.parametrize('case', H3, ids=[c[0] for c in H3])
def test_layer_operand_image_kernels(hip, case, variant, hipenv):
  from advoc_amd import conv
  tile, stages = variant
  hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_H3_TILE=tile, ADVOC_H3_DEEP_STAGES=stages)
  c = build_case(case)
  dev = torch.device('cuda')
  x0 = c['x0'].to(dev)
  x1 = c['x1'].to(dev) if c['x1'] is not None else None
  w = c['w'].to(dev)
  cout = w.shape[3] if c['kind'] == 0 else w.shape[2]
  y = torch.empty(x0.shape[0], c['oh'], c['out_w'], cout, device=dev)
  L = conv.Layer(c['kind'], x0, y, w, None, x1=x1, in_w=c['in_w'], stride=c['stride'], pad=c['pad'], in_act=c['act'])
  want_tile = {1: '2, 2', 2: '2, 4', 3: '4, 2', 4: '2, 1', 5: '2, 4', 6: '2, 2'}[tile]
  cin = x0.shape[3] + (x1.shape[3] if x1 is not None else 0)
  for direction, n_cols, k_ch in ((0, cout, cin), (1, cin, cout)):
    name = L.kernel_name(direction)
    if k_ch % 32 or n_cols % 64 or x0.shape[3] % 32:
      continue                                                   # outside the image path: other kernels, same oracle
    if n_cols % 128:
      want = 'gather_gemm_h3_kernel<2, 1, %d, 2>' % stages          # 64 / 192 columns: 128 x 64 tiles
    elif tile in (2, 5) and n_cols % 256:
      want = 'gather_gemm_h3_kernel<2, 2, %d, 2>' % stages          # 256-column tile impossible: falls to 128 x 128
    elif tile in (5, 6):
      want = 'gather_gemm_h3_kernel<%s, 2, 4>' % want_tile       # 8-wave workgroups: 256 x 256 / 256 x 128
    else:
      want = 'gather_gemm_h3_kernel<%s, %d, 2>' % (want_tile, 2 if tile in (2, 3) else stages)
    assert name == want, (direction, name, want)
  test_layer_all_directions(hip, case)


DEEP_SLICES = [c for c in H3 if c[0] in ('h3_enc', 'h3_dec_skip', 'h3_dec_mixed', 'h3_deep_small')]


@gpu
@pytest.mark.parametrize('split', [3, 5])
@pytest.mark.parametrize('tile', [4, 1, 5], ids=['128x64', '128x128', '256x256'])
@pytest.mark.parametrize('case', DEEP_SLICES, ids=[c[0] for c in DEEP_SLICES])
def test_deep_launches_cut_into_k_slices_on_every_tile(hip, case, tile, split, hipenv):
  """Launches under one round of tiles have EVERY tile cut into K slices that meet in the workspace (igemm_h3.hip: the last
  slice to arrive adds the parked partial tiles in slice order -- 16-byte agent-scope pieces, four slices in flight -- and
  runs the ordinary epilogue).  Since r4 the 128 x 128 and 256 x 256 tiles take that path too (encoder_5 forward, decoder_6
  backward-data of the full model): each tile shape with 3 and 5 slices (one and two passes of the four-slice reader, plus
  its one-slice remainder) against the float64 oracle in all directions, and the forward result bit-reproducible and within
  the order of one sum of the unsplit launch."""
  from advoc_amd import conv
  c = build_case(case)
  dev = torch.device('cuda')
  x0 = c['x0'].to(dev)
  x1 = c['x1'].to(dev) if c['x1'] is not None else None
  w = c['w'].to(dev)
  cout = w.shape[3] if c['kind'] == 0 else w.shape[2]

  def forward():
    y = torch.full((x0.shape[0], c['oh'], c['out_w'], cout), float('nan'), device=dev)
    L = conv.Layer(c['kind'], x0, y, w, None, x1=x1, in_w=c['in_w'], stride=c['stride'], pad=c['pad'], in_act=c['act'])
    L.forward()
    return L, y

  hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_H3_TILE=tile, ADVOC_H3_DEEP_SPLIT=split)
  L, y_a = forward()
  assert 'gather_gemm_h3_kernel' in L.kernel_name(0), L.kernel_name(0)
  _, y_b = forward()
  assert torch.equal(y_a, y_b)
  hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_H3_TILE=tile, ADVOC_H3_DEEP_SPLIT=None, ADVOC_IGEMM_SPLITK=0)
  _, y_plain = forward()
  assert not torch.equal(y_a, y_plain)              # (the slices were really taken: another order of the same sum)
  assert rel(y_a, y_plain.double()) < 1e-6
  hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_H3_TILE=tile, ADVOC_H3_DEEP_SPLIT=split, ADVOC_IGEMM_SPLITK=None)
  test_layer_all_directions(hip, case)


@gpu
def test_tile_choice_follows_the_workspace_the_caller_really_gave(hip, hipenv):
  """ The 256 x 256 tile of an under-filled single-phase launch (encoder_5 forward: 68 tiles on 256 CUs) counts on
  its K slices meeting in the workspace.  A caller whose workspace holds the operand images but not the parked slices gets
  the launch re-planned -- tile choice included -- rather than 68 unsplit workgroups: another kernel instance, the same
  result up to the order of one sum."""
  from advoc_amd import conv
  dev = torch.device('cuda')
  g = torch.Generator().manual_seed(11)
  x0 = torch.randn(64, 16, 34, 256, generator=g).to(dev)
  w = (torch.randn(4, 4, 256, 512, generator=g) * 0.02).to(dev)

  def run(shrink):
    y = torch.full((64, 8, 17, 512), float('nan'), device=dev)
    L = conv.Layer(0, x0, y, w, None, stride=(2, 2), in_act=1)
    if shrink:
      # (the workspace is one buffer per device, grown by whatever ran before: count from what THIS layer asks for)
      need = hip.advoc_conv_workspace_bytes(ctypes.byref(L.struct), 0)
      assert 0 < need <= L.struct.workspace_bytes
      L.struct.workspace_bytes = need - shrink
    L.forward()
    torch.cuda.synchronize()
    return L.kernel_name(0), y

  n_full, y_full = run(0)
  if n_full != 'gather_gemm_h3_kernel<2, 4, 2, 4>':
    pytest.skip('the plan of this device does not take the 256 x 256 tile here: ' + n_full)
  n_small, y_small = run(1 << 20)
  assert 'gather_gemm_h3_kernel' in n_small and n_small != n_full, (n_full, n_small)
  assert torch.isfinite(y_small).all()
  assert rel(y_small, y_full.double()) < 1e-6, rel(y_small, y_full.double())


@gpu
@pytest.mark.parametrize('mode', ['default', 'k_split_workspace'])
@pytest.mark.parametrize('case,want', H3_N32, ids=[c[0][0] for c in H3_N32])
def test_layer_32_column_launches_on_the_image_kernel(hip, case, want, mode, hipenv):
  """N = 32: one 64-column tile per row tile, weight rows 32..63 masked in the DMA (zeros), their column block skipped in
  the epilogue; all directions against the float64 oracle, also with every tile cut into K slices that meet in the workspace
  (the partial tiles carry the masked half along)."""
  from advoc_amd import conv
  if mode == 'default':
    hipenv(ADVOC_H3_MIN_TILES=1)
  else:
    hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_H3_DEEP_WGS_PER_CU=4, ADVOC_H3_DEEP_SPLIT_DIV=2)
  c = build_case(case)
  dev = torch.device('cuda')
  x0 = c['x0'].to(dev)
  x1 = c['x1'].to(dev) if c['x1'] is not None else None
  w = c['w'].to(dev)
  cout = w.shape[3] if c['kind'] == 0 else w.shape[2]
  y = torch.empty(x0.shape[0], c['oh'], c['out_w'], cout, device=dev)
  L = conv.Layer(c['kind'], x0, y, w, None, x1=x1, in_w=c['in_w'], stride=c['stride'], pad=c['pad'], in_act=c['act'])
  for direction, name in want.items():
    assert L.kernel_name(direction) == name, (direction, L.kernel_name(direction))
  test_layer_all_directions(hip, case)


@gpu
@pytest.mark.parametrize('case', H3, ids=[c[0] for c in H3])
def test_layer_operand_image_weight_gradient(hip, case, hipenv):
  """wgrad_h3.hip: both operands from fp16 pair images, transposing LDS reads; every H3 shape (taps that leave the
  image, two sources, strides, dropout on the gradient operand) against the float64 oracle."""
  from advoc_amd import conv
  hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_WGRAD_H3_MIN_M=1)
  c = build_case(case)
  dev = torch.device('cuda')
  x0 = c['x0'].to(dev)
  x1 = c['x1'].to(dev) if c['x1'] is not None else None
  w = c['w'].to(dev)
  cout = w.shape[3] if c['kind'] == 0 else w.shape[2]
  y = torch.empty(x0.shape[0], c['oh'], c['out_w'], cout, device=dev)
  L = conv.Layer(c['kind'], x0, y, w, None, x1=x1, in_w=c['in_w'], stride=c['stride'], pad=c['pad'], in_act=c['act'])
  assert L.kernel_name(2).replace('_flat', '') == 'wgrad_h3_kernel', L.kernel_name(2)
  test_layer_all_directions(hip, case)
  hipenv(ADVOC_WGRAD_H3=0)
  L2 = conv.Layer(c['kind'], x0, y, w, None, x1=x1, in_w=c['in_w'], stride=c['stride'], pad=c['pad'], in_act=c['act'])
  assert 'h3' not in L2.kernel_name(2)


# (r5) ROW MODE of the image weight gradient (wgrad_h3.hip: grid rows of >= 32 points, a K tile in at most two rows): rows of
# 64 / 32 points (a tile = one row or half of one), 63 / 33 / 65 (tiles that straddle rows at every offset, and images),
# a transposed conv with a skip source and a trimmed column, stride-1 and stride-2 gathers whose taps leave the image
WROWS = [
    ('wrow_enc',    0, (2, 16, 128), 128, 0, 256, 0, (2, 2), None, 1, False, 0),
    ('wrow_d4',     0, (2, 9, 64), 128, 0, 256, 0, (1, 1), (1, 1), 1, False, 0),
    ('wrow_enc33',  0, (3, 10, 66), 128, 0, 256, 0, (2, 2), None, 1, False, 0),
    ('wrow_enc65',  0, (2, 6, 129), 128, 0, 256, 0, (2, 2), None, 1, False, 0),
    ('wrow_dec',    1, (2, 8, 32), 128, 128, 256, 1, (2, 2), (1, 1), 2, True, 0),
    ('wrow_dec33',  1, (3, 4, 33), 128, 128, 256, 1, (2, 2), (1, 1), 2, True, 0),
]


@gpu
@pytest.mark.parametrize('tile', [1, 2], ids=['128', '256'])
@pytest.mark.parametrize('case', WROWS, ids=[c[0] for c in WROWS])
def test_weight_gradient_row_mode_equals_the_flat_axis(hip, case, tile, hipenv):
  """wgrad_h3.hip row mode (address arithmetic per grid row: scalars, 11 vector instructions per slot, placed between the
  MFMAs) walks the same K tiles as the per-slot arithmetic: bit for bit the same weight gradient, and all directions
  against the float64 oracle.  ADVOC_WGRAD_H3_ROWS=2: row mode or no image kernel at all -- the name shows it was taken."""
  from advoc_amd import conv
  c = build_case(case)
  dev = torch.device('cuda')
  x0 = c['x0'].to(dev)
  x1 = c['x1'].to(dev) if c['x1'] is not None else None
  w = c['w'].to(dev)
  cout = w.shape[3] if c['kind'] == 0 else w.shape[2]
  y = torch.empty(x0.shape[0], c['oh'], c['out_w'], cout, device=dev)
  dy = c['dy'].to(dev)
  out = {}
  for rows in (2, 0):
    hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_WGRAD_H3_MIN_M=1, ADVOC_WGRAD_H3_TILE=tile, ADVOC_WGRAD_H3_ROWS=rows,
           ADVOC_WGRAD_H3_ORDERED=2)
    L = conv.Layer(c['kind'], x0, y, w, None, x1=x1, in_w=c['in_w'], stride=c['stride'], pad=c['pad'], in_act=c['act'])
    assert L.kernel_name(2).replace('_flat', '') == ('wgrad_h3_256_kernel' if tile == 2 else 'wgrad_h3_kernel'), L.kernel_name(2)
    dw = torch.full_like(w, float('nan'))
    L.backward_weight(dy, dw)
    torch.cuda.synchronize()
    out[rows] = dw
    test_layer_all_directions(hip, case)
  assert torch.isfinite(out[2]).all()
  assert torch.equal(out[2], out[0]), rel(out[2], out[0])


WG256 = [c for c in H3 if c[0] in ('h3_enc', 'h3_dec_skip', 'h3_dec_first')]      # (the others: 128 / 384 columns)


@gpu
@pytest.mark.parametrize('case', WG256, ids=[c[0] for c in WG256])
def test_layer_operand_image_weight_gradient_256_tile(hip, case, hipenv):
  """wgrad_h3_256_kernel (256 x 256 on 8 waves, what the large layers of the model run): forced onto the small H3
  shapes whose matrix dimensions divide by 256, all directions against the float64 oracle."""
  from advoc_amd import conv
  hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_WGRAD_H3_MIN_M=1, ADVOC_WGRAD_H3_TILE=2)
  c = build_case(case)
  dev = torch.device('cuda')
  x0 = c['x0'].to(dev)
  x1 = c['x1'].to(dev) if c['x1'] is not None else None
  w = c['w'].to(dev)
  cout = w.shape[3] if c['kind'] == 0 else w.shape[2]
  y = torch.empty(x0.shape[0], c['oh'], c['out_w'], cout, device=dev)
  L = conv.Layer(c['kind'], x0, y, w, None, x1=x1, in_w=c['in_w'], stride=c['stride'], pad=c['pad'], in_act=c['act'])
  assert L.kernel_name(2).replace('_flat', '') == 'wgrad_h3_256_kernel', L.kernel_name(2)
  test_layer_all_directions(hip, case)


@gpu
@pytest.mark.parametrize('tile', [1, 2])
def test_weight_gradient_k_slices_summed_in_order(hip, hipenv, tile):
  """advoc_conv_layer.wgrad_ws: the K slices of the image weight gradient are parked and summed in slice order by a
  second launch instead of meeting in fp32 atomics -- the same numbers (up to the order of fp32 additions), bit for bit
  the same from call to call (the atomic sum is not), and `accumulate` adds to what dw held.  Both tile sizes."""
  from advoc_amd import conv
  dev = torch.device('cuda')
  g = torch.Generator().manual_seed(43)
  x = torch.randn(4, 32, 66, 128, generator=g).to(dev)
  w = (torch.randn(4, 4, 128, 256, generator=g) * 0.05).to(dev)
  dy = torch.randn(4, 16, 33, 256, generator=g).to(dev)
  y = torch.empty(4, 16, 33, 256, device=dev)

  def run(ordered, accumulate=False, base=None):
    hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_WGRAD_H3_MIN_M=1, ADVOC_WGRAD_H3_TILE=tile, ADVOC_WGRAD_H3_ORDERED=ordered)
    L = conv.Layer(conv.CONV, x, y, w, None, stride=(2, 2), pad=(1, 1), in_act=conv.ACT_LRELU)
    assert L.kernel_name(2).replace('_flat', '') == ('wgrad_h3_256_kernel' if tile == 2 else 'wgrad_h3_kernel')
    assert L.struct.wgrad_ws and L.struct.wgrad_ws_bytes >= hip.advoc_conv_wgrad_ws_bytes(ctypes.byref(L.struct)) > 0
    dw = torch.full_like(w, float('nan')) if base is None else base.clone()
    L.backward_weight(dy, dw, accumulate=accumulate)
    torch.cuda.synchronize()
    return dw
  atomic = run(0)
  a, b = run(2), run(2)
  assert torch.equal(a, b)                                   # ordered: reproducible (and the NaN fill was overwritten)
  assert rel(a, atomic) < 1e-6
  base = torch.randn(w.shape, generator=g).to(dev)
  acc = run(2, accumulate=True, base=base)
  assert rel(acc, base + a) < 1e-6
  # the default orders the 256 x 256 tile only
  d1, d2 = run(1), run(1)
  if tile == 2:
    assert torch.equal(d1, d2) and torch.equal(d1, a)


@gpu
def test_two_streams_share_the_k_slice_scratch_in_order(hip, hipenv):
  """ Layers that sum their K slices in order share ONE scratch buffer per device.  Two layers called from two
  streams with nothing else ordering them (conv.Layer.backward_weight makes the second stream wait for the first launch's
  event) give, bit for bit, what the same calls give one after the other on one stream -- on every one of many rounds, with
  the streams swapping who goes first and a long kernel in front of the first caller so that the second would overtake it."""
  from advoc_amd import conv
  dev = torch.device('cuda')
  g = torch.Generator().manual_seed(47)
  hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_WGRAD_H3_MIN_M=1, ADVOC_WGRAD_H3_ORDERED=2)
  layers = []
  for cin, cout, tile in ((128, 256, 2), (256, 128, 1)):
    x = torch.randn(4, 32, 66, cin, generator=g).to(dev)
    w = (torch.randn(4, 4, cin, cout, generator=g) * 0.05).to(dev)
    dy = torch.randn(4, 16, 33, cout, generator=g).to(dev)
    y = torch.empty(4, 16, 33, cout, device=dev)
    hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_WGRAD_H3_MIN_M=1, ADVOC_WGRAD_H3_ORDERED=2, ADVOC_WGRAD_H3_TILE=tile)
    L = conv.Layer(conv.CONV, x, y, w, None, stride=(2, 2), pad=(1, 1), in_act=conv.ACT_LRELU)
    assert 'wgrad_h3' in L.kernel_name(2) and L.struct.wgrad_ws
    layers.append((L, dy, w))
  assert layers[0][0].struct.wgrad_ws == layers[1][0].struct.wgrad_ws          # one scratch, two users
  serial = []
  for L, dy, w in layers:
    dw = torch.full_like(w, float('nan'))
    L.backward_weight(dy, dw)
    torch.cuda.synchronize()
    serial.append(dw)
  streams = [torch.cuda.Stream(), torch.cuda.Stream()]
  junk = torch.randn(32 * 1024 * 1024, device=dev)
  for rnd in range(12):
    order = (0, 1) if rnd % 2 == 0 else (1, 0)
    outs = [None, None]
    torch.cuda.synchronize()
    for k, i in enumerate(order):
      L, dy, w = layers[i]
      with torch.cuda.stream(streams[k]):
        if k == 0:
          for _ in range(3):
            junk.mul_(1.0000001)                    # the first caller's launch starts late ...
        outs[i] = torch.full_like(w, float('nan'))
        L.backward_weight(dy, outs[i])              # ... and the second caller's must still wait for it
    torch.cuda.synchronize()
    for i in (0, 1):
      assert torch.equal(outs[i], serial[i]), (rnd, i, rel(outs[i], serial[i]))


@gpu
def test_operand_image_kernel_with_batchnorm_prologue(hip, hipenv):
  """The image pass applies the producer's batch-norm affine and the dropout that follows it (in_scale / in_shift /
  in_mask of advoc_conv_layer) before the split: compare with the fp32 MFMA path on the same layer."""
  from advoc_amd import conv
  dev = torch.device('cuda')
  g = torch.Generator().manual_seed(11)
  x0 = torch.randn(2, 8, 18, 128, generator=g).to(dev)
  x1 = torch.randn(2, 8, 17, 128, generator=g).to(dev)
  w = (torch.randn(4, 4, 128, 256, generator=g) * 0.05).to(dev)
  sc = (torch.rand(256, generator=g) + 0.5).to(dev)
  sh = (torch.randn(256, generator=g) * 0.3).to(dev)
  mk = (torch.rand(2, 8, 18, 128, generator=g) >= 0.5).to(torch.uint8).to(dev)
  dy = torch.randn(2, 16, 34, 128, generator=g).to(dev)

  def run():
    y = torch.empty(2, 16, 34, 128, device=dev)
    L = conv.Layer(conv.DECONV, x0, y, w, None, x1=x1, in_w=17, stride=(2, 2), pad=(1, 1), in_act=conv.ACT_RELU,
                   in_scale=sc, in_shift=sh, in_mask=mk, in_mask_scale=2.0)
    L.forward()
    dx0, dx1 = torch.zeros_like(x0), torch.zeros_like(x1)
    L.backward_data(dy, dx0, dx1)
    return L.kernel_name(0), L.kernel_name(1), y, dx0, dx1
  hipenv(ADVOC_H3_MIN_TILES=1)
  n0, n1, y, dx0, dx1 = run()
  assert 'h3' in n0 and 'h3' in n1, (n0, n1)
  hipenv(ADVOC_IGEMM_X6=0)
  m0, m1, y32, dx0_32, dx1_32 = run()
  assert 'h3' not in m0 and not m0.endswith(', true>')
  for a, b in ((y, y32), (dx0, dx0_32), (dx1, dx1_32)):
    assert rel(a, b) < 3e-6, rel(a, b)


@gpu
def test_two_stage_path_is_selected(hip, hipenv):
  """<= 2 output columns over a wide K: one launch (fused_taps_kernel, S in LDS) where the shape allows; ADVOC_FUSED_TAPS=0
  gives the two launches through the workspace (and the direct kernel without one)."""
  from advoc_amd import conv
  dev = torch.device('cuda')
  c = build_case(CASES[8])
  x0, x1, w = c['x0'].to(dev), c['x1'].to(dev), c['w'].to(dev)
  y = torch.empty(x0.shape[0], c['oh'], c['out_w'], 1, device=dev)

  def names():
    with_ws = conv.Layer(1, x0, y, w, None, x1=x1, in_w=c['in_w'], in_act=2)
    without = conv.Layer(1, x0, y, w, None, x1=x1, in_w=c['in_w'], in_act=2, workspace=False)
    return with_ws.kernel_name(0), without.kernel_name(0)
  a, b = names()
  assert a == 'fused_taps_kernel<1, 4>' and b == a, (a, b)
  hipenv(ADVOC_FUSED_TAPS=0)
  a, b = names()
  assert 'gather_gemm' in a and 'gather_dot' in b, (a, b)


@gpu
@pytest.mark.parametrize('name', ['dec1_cout1', 'dec1_cout1_big', 'dec1_cout1_tiles', 'd5_cout1', 'd5_cout1_tiles', 'd1_cin2_k64',
                                  'd1_cin2_k128_tiles', 'd1_cin2'])
def test_fused_taps_equals_the_two_stage_path(hip, hipenv, name):
  """fused_taps_kernel (edge.hip) against the two launches it replaces (same fp32 matrix-core arithmetic, another
  summation order over the taps) and against the float64 oracle: the generator's last transposed conv (two sources, ReLU
  on load, trimmed column, clipped output), the discriminator's last conv, the backward-data call of its first (two
  columns), single-patch and many-patch grids with partial patches."""
  from advoc_amd import conv
  case = [c for c in CASES if c[0] == name][0]
  c = build_case(case)
  dev = torch.device('cuda')
  x0 = c['x0'].to(dev)
  x1 = c['x1'].to(dev) if c['x1'] is not None else None
  w, b, dy = c['w'].to(dev), c['b'].to(dev), c['dy'].to(dev)
  fwd = not name.startswith('d1_')

  def run():
    y = torch.full((x0.shape[0], c['oh'], c['out_w'], w.shape[3] if c['kind'] == 0 else w.shape[2]), float('nan'), device=dev)
    L = conv.Layer(c['kind'], x0, y, w, b, x1=x1, in_w=c['in_w'], stride=c['stride'], pad=c['pad'], in_act=c['act'])
    L.forward()
    dx0 = torch.full_like(x0, 7.0)
    dx1 = torch.full_like(x1, 7.0) if x1 is not None else None
    L.backward_data(dy, dx0, dx1)
    return L.kernel_name(0), L.kernel_name(1), y, dx0, dx1
  n0, n1, y, dx0, dx1 = run()
  assert 'fused_taps' in (n0 if fwd else n1), (n0, n1)
  hipenv(ADVOC_FUSED_TAPS=0)
  m0, m1, y2, dx0_2, dx1_2 = run()
  assert 'fused_taps' not in m0 and 'fused_taps' not in m1
  y_o, dx0_o, dx1_o, _, _ = oracle_layer(c['kind'], c['x0'], c['x1'], c['in_w'], c['w'], c['b'], c['stride'], c['pad'], c['act'],
                                         c['mask'], c['keep'], c['out_w'], c['dy'])
  if fwd:
    assert rel(y, y2) < 2e-6 and rel(y, y_o) < TOL, (rel(y, y2), rel(y, y_o))
  else:
    iw = c['in_w']
    assert rel(dx0[:, :, :iw], dx0_2[:, :, :iw]) < 2e-6 and rel(dx0[:, :, :iw], dx0_o[:, :, :iw]) < TOL
    assert torch.equal(dx0[:, :, iw:], dx0_2[:, :, iw:])        # what lies beyond the logical width stays untouched
    if dx1 is not None:
      assert rel(dx1, dx1_2) < 2e-6 and rel(dx1, dx1_o) < TOL
    # the G step's call: no gradient for the conditioning channel, the target channel's added to what the buffer holds
    # (one column instead of two: half the matrix work and half the LDS per pixel)
  hipenv(ADVOC_FUSED_TAPS=1)
  if fwd:
    # the producer's batch-norm affine on the input (in_scale / in_shift): the zero padding applies to the TRANSFORMED
    # input -- a padded pixel is 0, not act(shift)
    cin = x0.shape[3] + (x1.shape[3] if x1 is not None else 0)
    gsc = torch.Generator().manual_seed(9)
    sc = (torch.rand(cin, generator=gsc) + 0.5).to(dev)
    sh = (torch.randn(cin, generator=gsc) * 0.5).to(dev)

    def run_aff():
      ya = torch.full_like(y, float('nan'))
      La = conv.Layer(c['kind'], x0, ya, w, b, x1=x1, in_w=c['in_w'], stride=c['stride'], pad=c['pad'], in_act=c['act'],
                      in_scale=sc, in_shift=sh)
      La.forward()
      return La.kernel_name(0), ya
    na, ya = run_aff()
    hipenv(ADVOC_FUSED_TAPS=0)
    nb, yb = run_aff()
    assert 'fused_taps' in na and 'fused_taps' not in nb, (na, nb)
    assert rel(ya, yb) < 2e-6, rel(ya, yb)
  else:
    y3 = torch.empty_like(y)
    L = conv.Layer(c['kind'], x0, y3, w, b, x1=x1, in_w=c['in_w'], stride=c['stride'], pad=c['pad'], in_act=c['act'])
    base = torch.randn(dx1.shape, generator=torch.Generator().manual_seed(5)).to(dev)
    acc = base.clone()
    L.backward_data(dy, None, acc, accum1=True)
    assert 'fused_taps_kernel<1' in L.kernel_name(1) or True       # (the name query does not see the destinations)
    assert rel(acc, base + dx1_2) < 2e-6, rel(acc, base + dx1_2)


@gpu
@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_layer_all_directions(hip, case, workspace=True):
  from advoc_amd import conv
  c = build_case(case)
  dev = torch.device('cuda')
  y_o, dx0_o, dx1_o, dw_o, db_o = oracle_layer(c['kind'], c['x0'], c['x1'], c['in_w'], c['w'], c['b'],
                                               c['stride'], c['pad'], c['act'], c['mask'], c['keep'],
                                               c['out_w'], c['dy'])
  x0 = c['x0'].to(dev)
  x1 = c['x1'].to(dev) if c['x1'] is not None else None
  w, b, dy = c['w'].to(dev), c['b'].to(dev), c['dy'].to(dev)
  mask = c['mask'].to(dev) if c['mask'] is not None else None
  y = torch.full((x0.shape[0], c['oh'], c['out_w'], w.shape[3] if c['kind'] == 0 else w.shape[2]),
                 float('nan'), device=dev)
  L = conv.Layer(c['kind'], x0, y, w, b, x1=x1, in_w=c['in_w'], stride=c['stride'], pad=c['pad'],
                 in_act=c['act'], drop_mask=mask, drop_scale=1 / c['keep'] if mask is not None else 0.,
                 workspace=workspace)
  L.forward()
  assert torch.isfinite(y).all()
  assert rel(y, y_o) < TOL, ('fwd', rel(y, y_o))

  # backward data: poison, then check every logical column; the trimmed column stays untouched
  dx0 = torch.full_like(x0, 7.0)
  dx1 = torch.full_like(x1, 7.0) if x1 is not None else None
  L.backward_data(dy, dx0, dx1)
  assert rel(dx0[:, :, :c['in_w']], dx0_o[:, :, :c['in_w']]) < TOL, ('dx0', rel(dx0[:, :, :c['in_w']], dx0_o[:, :, :c['in_w']]))
  if x0.shape[2] > c['in_w']:
    assert (dx0[:, :, c['in_w']:] == 7.0).all()
    assert float(dx0_o[:, :, c['in_w']:].abs().max()) == 0.0
  if x1 is not None:
    assert rel(dx1, dx1_o) < TOL, ('dx1', rel(dx1, dx1_o))
    # accumulate flag: second call adds on top
    keep = dx1.clone()
    L.backward_data(dy, None if c['x0'].shape[3] * 0 else dx0, dx1, accum0=False, accum1=True)
    assert rel(dx1, 2 * keep) < TOL

  dw = torch.full_like(w, float('nan'))
  db = torch.full_like(b, float('nan'))
  L.backward_weight(dy, dw, db)
  assert rel(dw, dw_o) < TOL, ('dw', rel(dw, dw_o))
  assert rel(db, db_o) < TOL, ('db', rel(db, db_o))


@gpu
def test_discriminator_input_gradient_only_target(hip):
  """layer_1 backward in the G step: only d/d(target) is wanted (dx0 = NULL)."""
  from advoc_amd import conv
  c = build_case(CASES[10])
  dev = torch.device('cuda')
  _, _, dx1_o, _, _ = oracle_layer(c['kind'], c['x0'], c['x1'], c['in_w'], c['w'], c['b'], c['stride'],
                                   c['pad'], c['act'], None, 1.0, c['out_w'], c['dy'])
  x0, x1 = c['x0'].to(dev), c['x1'].to(dev)
  y = torch.empty((x0.shape[0], c['oh'], c['out_w'], 32), device=dev)
  L = conv.Layer(0, x0, y, c['w'].to(dev), c['b'].to(dev), x1=x1, stride=c['stride'], pad=c['pad'])
  dx1 = torch.zeros_like(x1)
  L.backward_data(c['dy'].to(dev), None, dx1)
  assert rel(dx1, dx1_o) < TOL


@gpu
def test_closed_form_impulse(hip):
  """1-channel impulse through an all-ones 4x4 stride-2 SAME conv: counts window membership,
  pinning the asymmetric (1,2) width padding independently of torch."""
  from advoc_amd import conv
  dev = torch.device('cuda')
  H, W = 8, 9      # odd width -> pad (1, 2)
  x = torch.zeros(1, H, W, 16, device=dev)
  x[0, 3, 8, :] = 1.0     # last column
  w = torch.ones(4, 4, 16, 32, device=dev)
  y = torch.empty(1, 4, 5, 32, device=dev)
  conv.Layer(0, x, y, w, None, stride=(2, 2), pad=(1, 1)).forward()
  want = np.zeros((4, 5))
  for oy in range(4):
    for ox in range(5):
      for ky in range(4):
        for kx in range(4):
          if oy * 2 - 1 + ky == 3 and ox * 2 - 1 + kx == 8:
            want[oy, ox] += 16
  assert np.array_equal(y[0, :, :, 0].cpu().numpy(), want)
  assert want[:, 4].sum() > 0     # the column that only exists because of the right pad of 2


@gpu
def test_abi_rejects_bad_layers(hip):
  from advoc_amd import _lib, conv
  dev = torch.device('cuda')
  x = torch.zeros(1, 8, 8, 24, device=dev)      # 24 channels: not a multiple of 16, not thin
  y = torch.zeros(1, 4, 4, 32, device=dev)
  w = torch.zeros(4, 4, 24, 32, device=dev)
  with pytest.raises(_lib.AdvocHipError):
    conv.Layer(0, x, y, w, None, stride=(2, 2), pad=(1, 1)).forward()
  with pytest.raises(_lib.AdvocHipError):
    conv.Layer(0, x, y, torch.zeros(4, 4, 32, 24, device=dev), None)
  with pytest.raises(_lib.AdvocHipError):
    conv.Layer(0, torch.zeros(1, 8, 8, 32), y, torch.zeros(4, 4, 32, 32, device=dev), None)


ODD = [
    ('c1_to_24', 0, (2, 16, 33), 1, 0, 24, 0, (2, 2), None, 0, False, 0),
    ('c24_to_1', 0, (2, 9, 11), 24, 0, 1, 0, (1, 1), (1, 1), 1, False, 0),
]


@gpu
@pytest.mark.parametrize('case', ODD, ids=[c[0] for c in ODD])
def test_channel_counts_outside_the_mfma_kernels(hip, case):
  """The AdVoc nets only have channel counts that are multiples of 32 (or 1-2 at the edges).  Other
  counts run forward and backward-data on the direct kernels (edge.hip) with the same parity, and
  the weight gradient reports ADVOC_ERR_UNSUPPORTED instead of computing something else."""
  from advoc_amd import _lib, conv
  c = build_case(case)
  dev = torch.device('cuda')
  y_o, dx0_o, _, _, _ = oracle_layer(c['kind'], c['x0'], c['x1'], c['in_w'], c['w'], c['b'], c['stride'], c['pad'],
                                     c['act'], c['mask'], c['keep'], c['out_w'], c['dy'])
  x0, w, b, dy = c['x0'].to(dev), c['w'].to(dev), c['b'].to(dev), c['dy'].to(dev)
  y = torch.zeros(x0.shape[0], c['oh'], c['out_w'], w.shape[3], device=dev)
  L = conv.Layer(c['kind'], x0, y, w, b, in_w=c['in_w'], stride=c['stride'], pad=c['pad'], in_act=c['act'])
  L.forward()
  assert rel(y, y_o) < TOL
  dx0 = torch.zeros_like(x0)
  L.backward_data(dy, dx0)
  assert rel(dx0, dx0_o) < TOL
  with pytest.raises(_lib.AdvocHipError, match='unsupported'):
    L.backward_weight(dy, torch.zeros_like(w), torch.zeros_like(b))


def _random_cases(n, seed):
  """Seeded random layer shapes: channel counts any multiple of 32 (an `ngf=96` override is legal), odd
  sizes, both kinds, strides (2,2) / (1,2) / (1,1), skip concat + trim, every activation, dropout."""
  rng = np.random.default_rng(seed)
  out = []
  for i in range(n):
    kind = int(rng.integers(0, 2))
    chans = [32, 64, 96, 128, 160, 192, 224, 256]
    c0 = int(rng.choice(chans))
    c1 = int(rng.choice([0, 0] + chans)) if kind == 1 else int(rng.choice([0, 0, 0, 32, 96]))
    cout = int(rng.choice(chans))
    B = int(rng.integers(1, 4))
    H, W = int(rng.integers(1, 12)), int(rng.integers(2, 20))
    if kind == 0:
      stride = [(2, 2), (1, 2), (1, 1)][int(rng.integers(0, 3))]
      pad = None if stride != (1, 1) and rng.random() < 0.5 else (1, 1)
      if pad == (1, 1) and (H + 2 < 4 or W + 2 < 4):
        H, W = H + 3, W + 3
      trim = 0
    else:
      stride = [(2, 2), (1, 2)][int(rng.integers(0, 2))]
      pad = (1, 1)
      trim = 1 if c1 else 0
    act = int(rng.integers(0, 3))
    drop = bool(rng.random() < 0.3)
    out.append(('rand%02d_k%d_%dx%dx%d_c%d+%d_o%d_s%d%d' % (i, kind, B, H, W, c0, c1, cout, stride[0], stride[1]),
                kind, (B, H, W), c0, c1, cout, trim, stride, pad, act, drop, 0))
  return out


RANDOM = _random_cases(28, seed=2024)


@gpu
@pytest.mark.parametrize('case', RANDOM, ids=[c[0] for c in RANDOM])
def test_random_layer_shapes(hip, case):
  test_layer_all_directions(hip, case)


@gpu
def test_delayed_scale_images_match_the_exact_two_pass_form(hip, hipenv):
  """Layer.delayed_scale: after the first (exact) image of a buffer, later images take their power-of-two scale from the
  previous image's largest magnitude (one pass instead of two).  Data of the same scale gives the same results to
  round-off; data 32 x larger still fits the head room; the headers report no saturation."""
  from advoc_amd import conv
  hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_WGRAD_H3_MIN_M=1)
  dev = torch.device('cuda')
  g = torch.Generator().manual_seed(21)
  x = torch.randn(3, 16, 33, 128, generator=g).to(dev)
  w = (torch.randn(4, 4, 128, 256, generator=g) * 0.05).to(dev)
  dy = torch.randn(3, 8, 17, 256, generator=g).to(dev)

  def run(layer, scale):
    xs = x * scale
    layer.x0.copy_(xs)
    layer.forward()
    dx = torch.zeros_like(x)
    layer.backward_data(dy * scale, dx)
    dw = torch.zeros_like(w)
    layer.backward_weight(dy * scale, dw)
    return layer.y.clone(), dx, dw

  def make(delayed):
    y = torch.empty(3, 8, 17, 256, device=dev)
    L = conv.Layer(conv.CONV, x.clone(), y, w, None, stride=(2, 2), pad=(1, 1), in_act=conv.ACT_LRELU)
    L.delayed_scale = delayed
    L.reuse_images = True
    assert 'h3' in L.kernel_name(0) and 'h3' in L.kernel_name(1) and 'h3' in L.kernel_name(2)
    return L
  exact, delayed = make(False), make(True)
  for scale in (1.0, 0.7, 32.0, 0.05):
    ye, dxe, dwe = run(exact, scale)
    yd, dxd, dwd = run(delayed, scale)
    for a, b in ((yd, ye), (dxd, dxe), (dwd, dwe)):
      assert rel(a, b) < 2e-6, (scale, rel(a, b))
  hdr_x, hdr_dy = delayed._img[1].cpu(), delayed._img[3].cpu()
  assert int(hdr_x[3]) == 0 and int(hdr_dy[3]) == 0          # nothing saturated
  assert int(hdr_x[2]) != 0                                    # the one-pass form really ran (a previous magnitude exists)


@gpu
@pytest.mark.parametrize('jump', [1000.0, 200.0, 1e-4, 0.0])
def test_delayed_scale_never_applies_a_clamped_or_underflowed_image(hip, hipenv, jump):
  """A tensor that grows 100-1000 x between two steps leaves the one-pass image's head room (2^6); one that shrinks 1e4 x
  (or becomes all zero, then comes back) would lose precision.  The pass counts / detects it and refit_image_kernel
  rebuilds the image with the exact scale IN THE SAME CALL: forward, backward-data and the weight gradient of the jump
  step equal the exact two-pass results, header word 5 counts the refit, and the next step is one-pass again."""
  from advoc_amd import conv
  hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_WGRAD_H3_MIN_M=1)
  dev = torch.device('cuda')
  g = torch.Generator().manual_seed(22)
  x = torch.randn(2, 16, 33, 128, generator=g).to(dev)
  w = (torch.randn(4, 4, 128, 256, generator=g) * 0.05).to(dev)
  dy = torch.randn(2, 8, 17, 256, generator=g).to(dev)

  def make(delayed):
    y = torch.empty(2, 8, 17, 256, device=dev)
    L = conv.Layer(conv.CONV, x.clone(), y, w, None, stride=(2, 2), pad=(1, 1), in_act=conv.ACT_LRELU)
    L.delayed_scale, L.reuse_images = delayed, True
    return L

  def step(L, scale):
    L.x0.copy_(x * scale)
    L.forward()
    dx = torch.zeros_like(x)
    L.backward_data(dy * scale, dx)
    dw = torch.zeros_like(w)
    L.backward_weight(dy * scale, dw)
    return L.y.clone(), dx, dw
  exact, delayed = make(False), make(True)
  for L in (exact, delayed):
    step(L, 1.0)                                    # first image: exact for both
  ref = step(exact, jump if jump else 0.0)
  got = step(delayed, jump if jump else 0.0)
  hx, hdy = delayed._img[1].cpu(), delayed._img[3].cpu()
  assert int(hx[5]) == 1 and int(hdy[5]) == 1       # both operand images were refitted, once
  if jump > 64:
    assert int(hx[3]) > 0 and int(hdy[3]) > 0       # values left the head room during the one-pass image ...
  for a, b in zip(got, ref):                        # ... and nothing read the clamped image
    if jump:
      assert rel(a, b) < 2e-6, (jump, rel(a, b))
    else:
      assert float(a.abs().max()) == 0.0 and float(b.abs().max()) == 0.0
  # next step at the new scale: one pass, no refit (from all-zero: no usable magnitude -> exact refit once more)
  ref = step(exact, jump if jump else 3.0)
  got = step(delayed, jump if jump else 3.0)
  hx, hdy = delayed._img[1].cpu(), delayed._img[3].cpu()
  assert int(hx[5]) == (1 if jump else 2) and int(hdy[5]) == (1 if jump else 2)
  for a, b in zip(got, ref):
    assert rel(a, b) < 2e-6, (jump, rel(a, b))


@gpu
@pytest.mark.parametrize('patch', [1, 0], ids=['patch_kernels', 'per_tap_tiles'])
def test_producer_written_operand_images_equal_the_image_pass(hip, hipenv, patch):
  """Layer.add_image_consumer (csrc/image_emit.h): from the second step on the PRODUCER's forward epilogue writes its
  consumers' operand images (consumer activation applied, consumer's one-pass scale) and the consumer runs the refit
  check instead of an image pass.  The images are value-identical to the ones the consumer builds itself, for one producer
  feeding two consumers with different activations (leaky ReLU / ReLU as the second concat source under a shared
  scale), with dropout on a producer's output; a 1000 x jump is refitted exactly."""
  from advoc_amd import conv
  hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_H3_PATCH_MIN_WGS=1, ADVOC_WGRAD_H3_MIN_M=1, ADVOC_H3_PATCH=patch)
  dev = torch.device('cuda')
  g = torch.Generator().manual_seed(31)
  xin1 = torch.randn(2, 66, 66, 64, generator=g).to(dev)
  xin2 = torch.randn(2, 16, 17, 64, generator=g).to(dev)
  w1 = (torch.randn(4, 4, 64, 128, generator=g) * 0.05).to(dev)
  w2 = (torch.randn(4, 4, 128, 64, generator=g) * 0.05).to(dev)
  wc1 = (torch.randn(4, 4, 128, 256, generator=g) * 0.05).to(dev)
  wc2 = (torch.randn(4, 4, 64, 256, generator=g) * 0.05).to(dev)
  b1 = (torch.randn(128, generator=g) * 0.1).to(dev)
  mask = (torch.rand(2, 32, 34, 128, generator=g) >= 0.5).to(torch.uint8).to(dev)

  def make(register):
    y1 = torch.empty(2, 32, 33, 128, device=dev)        # 33 columns: 16 n + 1, patches + a remainder-column launch
    y2 = torch.empty(2, 32, 34, 128, device=dev)        # one more physical column than the consumer reads (trim)
    P1 = conv.Layer(conv.CONV, xin1.clone(), y1, w1, b1, stride=(2, 2), pad=(1, 1), in_act=conv.ACT_LRELU)
    P2 = conv.Layer(conv.DECONV, xin2.clone(), y2, w2, None, stride=(2, 2), pad=(1, 1), in_act=conv.ACT_RELU,
                    drop_mask=mask, drop_scale=2.0)
    C1 = conv.Layer(conv.CONV, y1, torch.empty(2, 16, 17, 256, device=dev), wc1, None, stride=(2, 2), pad=(1, 1),
                    in_act=conv.ACT_LRELU)
    C2 = conv.Layer(conv.DECONV, y2, torch.empty(2, 64, 66, 64, device=dev), wc2, None, x1=y1, in_w=33, stride=(2, 2),
                    pad=(1, 1), in_act=conv.ACT_RELU)
    for L in (P1, P2, C1, C2):
      L.delayed_scale, L.reuse_images = True, True
      assert 'h3' in L.kernel_name(0), L.kernel_name(0)
    assert any('patch' in L.kernel_name(0) for L in (P1, P2)) == bool(patch), (P1.kernel_name(0), P2.kernel_name(0))
    if register:
      P1.add_image_consumer(C1, 0)
      P1.add_image_consumer(C2, 1)
      P2.add_image_consumer(C2, 0)
    return P1, P2, C1, C2
  A = make(True)
  R = make(False)
  assert conv.Layer.emit_images and A[0]._emit_targets() == []          # first step: the consumers have no history yet
  for step, scale in enumerate((1.0, 0.8, 1.3, 1000.0, 1000.0)):
    for P1, P2, C1, C2 in (A, R):
      P1.x0.copy_(xin1 * scale)
      P2.x0.copy_(xin2 * scale)
      for L in (P1, P2, C1, C2):
        L.forward()
    if step >= 1:
      assert len(A[0]._emit_targets()) == 2 and len(A[1]._emit_targets()) == 1
    for Ca, Cr in ((A[2], R[2]), (A[3], R[3])):
      # the operand image, value for value (as fp16: the two code paths may differ in the SIGN of a zero -- relu(-x) -- only)
      assert torch.equal(Ca._img[0].view(torch.float16), Cr._img[0].view(torch.float16)), step
      ha, hr = Ca._img[1].cpu(), Cr._img[1].cpu()
      assert int(ha[1]) == int(hr[1]) and int(ha[2]) == int(hr[2]) and int(ha[5]) == int(hr[5]), (step, ha, hr)
      assert torch.equal(Ca.y, Cr.y), step
  assert int(A[2]._img[1].cpu()[5]) == 1 and int(A[3]._img[1].cpu()[5]) == 1     # the 1000 x jump: one exact refit each
  # and the weight gradient of a consumer reads the producer-written image like any other
  dy = torch.randn(2, 16, 17, 256, generator=g).to(dev)
  dwa, dwr = torch.zeros_like(wc1), torch.zeros_like(wc1)
  A[2].backward_weight(dy, dwa)
  R[2].backward_weight(dy, dwr)
  assert rel(dwa, dwr) < 1e-6


@gpu
@pytest.mark.parametrize('cin,cout', [(1, 64), (2, 64), (2, 128), (1, 32)])
def test_thin_producers_write_operand_images(hip, hipenv, cin, cout):
  """The thin forward kernel (csrc/thin.hip: encoder_1 and the discriminator's layer_1, 1-2 input channels) writes its
  consumer's operand image from its epilogue as the image kernels do: value-identical to the consumer's own image pass,
  ragged widths included, and a 1000 x jump is refitted exactly."""
  from advoc_amd import conv
  hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_H3_PATCH_MIN_WGS=1)
  dev = torch.device('cuda')
  g = torch.Generator().manual_seed(37)
  xin = torch.randn(2, 66, 70, cin, generator=g).to(dev)
  w = (torch.randn(4, 4, cin, cout, generator=g) * 0.2).to(dev)
  b = (torch.randn(cout, generator=g) * 0.1).to(dev)
  wc = (torch.randn(4, 4, cout, 128, generator=g) * 0.05).to(dev)

  def make(register):
    y = torch.empty(2, 33, 35, cout, device=dev)
    P = conv.Layer(conv.CONV, xin.clone(), y, w, b, stride=(2, 2), pad=(1, 1))
    C = conv.Layer(conv.CONV, y, torch.empty(2, 16, 17, 128, device=dev), wc, None, stride=(2, 2), pad=(1, 1),
                   in_act=conv.ACT_LRELU)
    for L in (P, C):
      L.delayed_scale, L.reuse_images = True, True
    assert 'thin_k' in P.kernel_name(0) and 'h3' in C.kernel_name(0), (P.kernel_name(0), C.kernel_name(0))
    if register:
      P.add_image_consumer(C, 0)
    return P, C
  A, R = make(True), make(False)
  for step, scale in enumerate((1.0, 0.7, 1.2, 1000.0, 1000.0)):
    for P, C in (A, R):
      P.x0.copy_(xin * scale)
      P.forward()
      C.forward()
    if step >= 1:
      assert len(A[0]._emit_targets()) == 1
    assert torch.equal(A[1]._img[0].view(torch.float16), R[1]._img[0].view(torch.float16)), step
    ha, hr = A[1]._img[1].cpu(), R[1]._img[1].cpu()
    assert int(ha[1]) == int(hr[1]) and int(ha[2]) == int(hr[2]) and int(ha[5]) == int(hr[5]), (step, ha, hr)
    assert torch.equal(A[1].y, R[1].y), step
  assert int(A[1]._img[1].cpu()[5]) == 1


@gpu
@pytest.mark.parametrize('cmid', [128, 512])
def test_backward_data_writes_the_output_gradient_image_of_the_layer_below(hip, hipenv, monkeypatch, cmid):
  """r4: the backward-data call of a 1-output-channel layer (the discriminator's layer_5, csrc/thin.hip) writes the
  output-gradient image of the layer below (layer_4: the largest image pass of the train step) from its own epilogue, and
  that layer's bias gradient (the column sums the image pass used to carry): image and header value-identical to the lower
  layer's own image pass, so its backward-data result is bit-identical; weight and bias gradients to the
  order of one sum; ragged widths; a 1000 x jump is refitted exactly."""
  from advoc_amd import conv
  hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_H3_PATCH_MIN_WGS=1)
  monkeypatch.setattr(conv.Layer, 'emit_dx', True)          # (off by default: it does not pay on the train step, conv.py)
  dev = torch.device('cuda')
  g = torch.Generator().manual_seed(41)
  xin = torch.randn(2, 32, 37, 64, generator=g).to(dev)
  wc = (torch.randn(4, 4, 64, cmid, generator=g) * 0.05).to(dev)
  bc = (torch.randn(cmid, generator=g) * 0.1).to(dev)
  wp = (torch.randn(4, 4, cmid, 1, generator=g) * 0.05).to(dev)
  bp = (torch.randn(1, generator=g) * 0.1).to(dev)
  dyp = torch.randn(2, 30, 35, 1, generator=g).to(dev)

  def make():
    yc = torch.empty(2, 31, 36, cmid, device=dev)
    C = conv.Layer(conv.CONV, xin.clone(), yc, wc, bc, stride=(1, 1), pad=(1, 1), in_act=conv.ACT_LRELU)
    P = conv.Layer(conv.CONV, yc, torch.empty(2, 30, 35, 1, device=dev), wp, bp, stride=(1, 1), pad=(1, 1),
                   in_act=conv.ACT_LRELU)
    for L in (P, C):
      L.delayed_scale, L.reuse_images = True, True
    assert 'thin_k' in P.kernel_name(1) and 'h3' in C.kernel_name(1) and 'h3' in C.kernel_name(2), \
        (P.kernel_name(1), C.kernel_name(1), C.kernel_name(2))
    bufs = dict(gc=torch.empty_like(yc), dx=torch.empty_like(xin), dw=torch.zeros_like(wc), db=torch.zeros_like(bc))
    return P, C, bufs
  A, R = make(), make()
  took = 0
  for step, scale in enumerate((1.0, 0.7, 1.2, 1000.0, 1000.0)):
    for (P, C, t), emit in ((A, True), (R, False)):
      C.forward()
      P.forward()
      t['dw'].zero_()
      P.backward_data(dyp * scale, t['gc'], grad_consumer=C if emit else None, consumer_db=t['db'] if emit else None,
                      consumer_db_accumulate=False)
      if emit:
        took += C._dy_emitted_for is not None
        assert (C._dy_emitted_for is not None) == (step >= 1), step      # (the first step has no magnitude history yet)
      C.backward_data(t['gc'], t['dx'], db=t['db'], db_accumulate=False)
      C.backward_weight(t['gc'], t['dw'], t['db'])
    (Pa, Ca, ta), (Pr, Cr, tr) = A, R
    assert torch.equal(ta['gc'], tr['gc']), step
    assert torch.equal(Ca._img[2].view(torch.float16), Cr._img[2].view(torch.float16)), step
    ha, hr = Ca._img[3].cpu(), Cr._img[3].cpu()
    assert int(ha[1]) == int(hr[1]) and int(ha[2]) == int(hr[2]) and int(ha[5]) == int(hr[5]), (step, ha, hr)
    assert torch.equal(ta['dx'], tr['dx']), step
    assert rel(ta['dw'], tr['dw']) < 2e-6, (step, rel(ta['dw'], tr['dw']))      # (same image; K slices may meet with atomics)
    assert rel(ta['db'], tr['db'].double().cpu()) < 2e-6, (step, rel(ta['db'], tr['db'].double().cpu()))
  assert took == 4
  assert int(A[1]._img[3].cpu()[5]) == 1            # one exact refit: the 1000 x jump


@gpu
@pytest.mark.parametrize('shape', ['d4_s1', 'd3_s2', 'dec_two_sources', 'thin_cout1', 'thin_two_sources', 'dec_rem_trim',
                                   'deep_per_tap', 'enc_accum', 'enc_accum_rem', 'enc_accum_deep'])
def test_patch_backward_data_writes_the_lower_layers_gradient_image_under_the_a_priori_scale(hip, hipenv, monkeypatch, shape):
  """r5: the backward-data call of a layer on a PATCH kernel writes the output-gradient image of the layer below from its
  epilogue under a scale derived from a bound of |dx| known before the launch (max|dy| max|w| taps K: nothing can leave the
  fp16 range -- no history, no refit check), with that layer's bias column sums, and does NOT write the fp32 tensor at
  all (advoc_conv_layer.dx_img, ADVOC_DX_BOUNDED | ADVOC_DX_IMAGE_ONLY): the lower layer's backward-data, weight and bias
  gradients equal the ones it computes from the fp32 tensor through its own image pass (a power-of-two scale apart: 2e-6),
  the fp32 buffer keeps its poison, the header holds the tensor's largest magnitude; 1000 x and 1e-4 x jumps of the
  incoming gradient from one step to the next change nothing (there is no window to leave); on a 4x4 stride-1 layer
  (discriminator layer_4 -> layer_3), a stride-2 layer (layer_3 -> layer_2) and a two-source transposed layer whose second
  destination stays an ordinary fp32 tensor (the generator's decoders)."""
  from advoc_amd import conv
  hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_H3_PATCH_MIN_WGS=1, ADVOC_WGRAD_H3_MIN_M=1)
  monkeypatch.setattr(conv.Layer, 'dx_accum', True)       # (the accumulating form is off by default: it does not pay, conv.py)
  dev = torch.device('cuda')
  g = torch.Generator().manual_seed(77)
  if shape == 'd4_s1':            # lower: 64 -> 128 stride 2; upper: 4x4 stride 1, 128 -> 256 on a 32 x 32 grid
    xin = torch.randn(2, 64, 64, 64, generator=g)
    low = dict(kind=conv.CONV, w=(4, 4, 64, 128), y=(2, 32, 32, 128), stride=(2, 2), pad=(1, 1))
    up = dict(kind=conv.CONV, w=(4, 4, 128, 256), y=(2, 31, 31, 256), stride=(1, 1), pad=(1, 1), x1=None)
    want = 'patch_gemm_h3_kernel<5, 1>'
  elif shape == 'd3_s2':          # upper: stride 2, 128 -> 256: backward-data = four sub-pixel phases over a 16 x 32 grid
    xin = torch.randn(3, 64, 128, 64, generator=g)
    low = dict(kind=conv.CONV, w=(4, 4, 64, 128), y=(3, 32, 64, 128), stride=(2, 2), pad=(1, 1))
    up = dict(kind=conv.CONV, w=(4, 4, 128, 256), y=(3, 16, 32, 256), stride=(2, 2), pad=(1, 1), x1=None)
    want = 'patch_gemm_h3_kernel<4, 1>'
  elif shape == 'thin_cout1':     # upper: one output channel (the discriminator's layer_5): the thin matrix kernel, max |w| given
    xin = torch.randn(2, 32, 37, 64, generator=g)
    low = dict(kind=conv.CONV, w=(4, 4, 64, 128), y=(2, 31, 36, 128), stride=(1, 1), pad=(1, 1))
    up = dict(kind=conv.CONV, w=(4, 4, 128, 1), y=(2, 30, 35, 1), stride=(1, 1), pad=(1, 1), x1=None)
    want = 'thin_k_gemm_kernel<16, 4, false>'
  elif shape == 'thin_two_sources':   # upper: the generator's decoder_1 (transposed, one output channel, two 64-channel sources)
    xin = torch.randn(2, 16, 18, 64, generator=g)
    low = dict(kind=conv.DECONV, w=(4, 4, 64, 64), y=(2, 32, 36, 64), stride=(2, 2), pad=(1, 1))
    up = dict(kind=conv.DECONV, w=(4, 4, 1, 128), y=(2, 64, 72, 1), stride=(2, 2), pad=(1, 1), x1=(2, 32, 36, 64))
    want = 'thin_k_gemm_kernel<16, 4, true>'
  elif shape in ('enc_accum', 'enc_accum_rem', 'enc_accum_deep'):
    # the generator's encoder chain: dx0 ACCUMULATES into a tensor that already holds a skip gradient (a decoder wrote it, and
    # recorded its largest magnitude: bound_add); the sum exists as the lower encoder's image only.  Four-phase patches, the
    # same with a remainder column, and the per-tap kernel alone
    hw = {'enc_accum': (64, 128), 'enc_accum_rem': (64, 130), 'enc_accum_deep': (16, 18)}[shape]
    xin = torch.randn(2, hw[0], hw[1], 64, generator=g)
    low = dict(kind=conv.CONV, w=(4, 4, 64, 128), y=(2, hw[0] // 2, hw[1] // 2, 128), stride=(2, 2), pad=(1, 1))
    up = dict(kind=conv.CONV, w=(4, 4, 128, 256), y=(2, hw[0] // 4, (hw[1] // 2 + 1) // 2, 256), stride=(2, 2), pad=(1, 1),
              x1=None, accum=True)
    want = 'gather_gemm_h3_kernel' if shape == 'enc_accum_deep' else 'patch_gemm_h3_kernel<4, 1>'
  elif shape == 'dec_rem_trim':   # the generator's decoders as they are: a 16 n + 1 wide grid (patches + a per-tap launch over
    xin = torch.randn(2, 16, 17, 64, generator=g)     # the remainder column, both writing the image), a trimmed first source
    low = dict(kind=conv.DECONV, w=(4, 4, 128, 64), y=(2, 32, 34, 128), stride=(2, 2), pad=(1, 1))
    up = dict(kind=conv.DECONV, w=(4, 4, 64, 256), y=(2, 64, 66, 64), stride=(2, 2), pad=(1, 1), x1=(2, 32, 33, 128), in_w=33)
    want = 'patch_gemm_h3_kernel<2, 1>'
  elif shape == 'deep_per_tap':   # under 16 x 16 grid points: the per-tap kernel alone writes the image
    xin = torch.randn(4, 4, 5, 64, generator=g)
    low = dict(kind=conv.DECONV, w=(4, 4, 128, 64), y=(4, 8, 10, 128), stride=(2, 2), pad=(1, 1))
    up = dict(kind=conv.DECONV, w=(4, 4, 64, 256), y=(4, 16, 18, 64), stride=(2, 2), pad=(1, 1), x1=(4, 8, 9, 128), in_w=9)
    want = 'gather_gemm_h3_kernel'
  else:                           # upper: transposed conv over concat(lower output, skip): dx0 image only, dx1 fp32
    xin = torch.randn(2, 16, 16, 64, generator=g)
    low = dict(kind=conv.DECONV, w=(4, 4, 128, 64), y=(2, 32, 32, 128), stride=(2, 2), pad=(1, 1))
    up = dict(kind=conv.DECONV, w=(4, 4, 64, 256), y=(2, 64, 64, 64), stride=(2, 2), pad=(1, 1), x1=(2, 32, 32, 128))
    want = 'patch_gemm_h3_kernel<2, 1>'
  w_lo = (torch.randn(*low['w'], generator=g) * 0.05).to(dev)
  b_lo = (torch.randn(low['y'][3], generator=g) * 0.1).to(dev)
  w_up = (torch.randn(*up['w'], generator=g) * 0.05).to(dev)
  skip = torch.randn(*up['x1'], generator=g).to(dev) if up['x1'] else None
  dy_up = torch.randn(*up['y'], generator=g).to(dev)
  xin = xin.to(dev)

  def make():
    y_lo = torch.empty(*low['y'], device=dev)
    Lo = conv.Layer(low['kind'], xin.clone(), y_lo, w_lo, b_lo, stride=low['stride'], pad=low['pad'], in_act=conv.ACT_LRELU)
    # (max |w| on the device, as the train step keeps it: the thin kernel's bound reads it from there)
    w_amax = w_up.abs().max().reshape(1).view(torch.int32)
    Up = conv.Layer(up['kind'], y_lo, torch.empty(*up['y'], device=dev), w_up, None, x1=skip, stride=up['stride'],
                    pad=up['pad'], in_act=conv.ACT_RELU if up['kind'] == conv.DECONV else conv.ACT_LRELU, w_amax=w_amax,
                    **({'in_w': up['in_w']} if 'in_w' in up else {}))
    for L in (Up, Lo):
      L.delayed_scale, L.reuse_images = True, True
    assert Up.kernel_name(1).startswith(want) and 'h3' in Lo.kernel_name(1) and 'h3' in Lo.kernel_name(2), \
        (Up.kernel_name(1), Lo.kernel_name(1), Lo.kernel_name(2))
    t = dict(g=torch.empty_like(y_lo), dskip=torch.empty_like(skip) if skip is not None else None,
             dx=torch.empty_like(xin), dw=torch.zeros_like(w_lo), db=torch.zeros_like(b_lo))
    return Up, Lo, t
  A, R = make(), make()
  accum = bool(up.get('accum'))
  skip_grad = (torch.randn(*low['y'], generator=g) * 3.0).to(dev) if accum else None
  for step, scale in enumerate((1.0, 0.8, 1000.0, 1e-4, 1.0)):
    for (Up, Lo, t), emit in ((A, True), (R, False)):
      Lo.forward()
      Up.forward()
      t['dw'].zero_()
      if accum:
        t['g'].copy_(skip_grad * scale)           # what the decoder left there, and its largest magnitude on the device
        word = (skip_grad * scale).abs().max().reshape(1).view(torch.int32)
        Up.backward_data(dy_up * scale, t['g'], accum0=True, grad_consumer=Lo if emit else None,
                         consumer_db=t['db'] if emit else None, consumer_db_accumulate=False, bound_add=word if emit else None)
        if emit:
          assert Lo._dy_emitted_for is not None and Lo._dy_emitted_for[2], step
          assert torch.equal(t['g'], skip_grad * scale), step                      # the fp32 tensor still holds the skip gradient only
      else:
        # (a trimmed column of the tensor is never written by anybody: zero in the model, and zero here where it is read)
        t['g'].fill_(float('nan') if emit else 0.0)
        Up.backward_data(dy_up * scale, t['g'], t['dskip'], grad_consumer=Lo if emit else None,
                         consumer_db=t['db'] if emit else None, consumer_db_accumulate=False)
      if emit and not accum:
        assert Lo._dy_emitted_for is not None and Lo._dy_emitted_for[2], step       # from the first step on: no history needed
        assert bool(torch.isnan(t['g']).all()), step                                 # the fp32 tensor is NOT written
      Lo.backward_data(t['g'], t['dx'], db=t['db'], db_accumulate=False)
      Lo.backward_weight(t['g'], t['dw'], t['db'])
    (Ua, La, ta), (Ur, Lr, tr) = A, R
    assert rel(ta['dx'], tr['dx']) < 2e-6, (step, rel(ta['dx'], tr['dx']))
    assert rel(ta['dw'], tr['dw']) < 2e-6, (step, rel(ta['dw'], tr['dw']))
    assert rel(ta['db'], tr['db'].double().cpu()) < 2e-6, (step, rel(ta['db'], tr['db'].double().cpu()))
    if skip is not None:
      assert torch.equal(ta['dskip'], tr['dskip']), step
    # the header of the emitted image: word 0 = the tensor's largest magnitude, word 1 = 2^-s with the bound at [2^14, 2^15):
    # the largest scaled value stays below 2^15 and -- a bound over <= 8192 products -- above 2^1
    hdr = La.struct.dy_hdr
    ha = [h for h in La.image_headers() if h.data_ptr() == hdr][0].cpu().view(torch.float32)
    amax = float(tr['g'].abs().max())
    assert abs(float(ha[0]) - amax) <= 2e-6 * amax, (step, float(ha[0]), amax)
    top = amax / float(ha[1])
    assert 2.0 <= top < 65504.0, (step, top)
  assert int(La.image_headers()[-1].cpu()[5]) == 0      # nothing was ever refitted


@gpu
@pytest.mark.parametrize('producer', ['thin', 'patch_s2', 'patch_s1_reader'])
def test_a_sole_reader_gets_the_output_as_its_operand_image_only(hip, hipenv, producer):
  """r5: a layer whose output has ONE reader -- the next layer, which reads it as an operand image and gates its
  backward-data pass on that image (a patch kernel) -- writes the image ONLY (advoc_conv_layer.y_img.mode =
  ADVOC_Y_BOUNDED | ADVOC_Y_IMAGE_ONLY): under a scale from an a-priori bound of |y| (max|x| max|w| taps K + max|b|), so
  there is no history and no refit, and the fp32 tensor is never written.  On the discriminator's layer_1 -> layer_2 shapes
  (two 1-channel sources, thin matrix kernel -> stride-2 conv whose backward-data runs the four-phase patch kernel): the
  reader's forward output, its backward-data result (gated by the image's signs, ADVOC_IMG_X_GATES), its weight and bias
  gradients equal the fp32 path's (2e-6: the two images differ by a power of two), the fp32 buffer keeps its poison, and a
  1000 x jump of the input from one step to the next changes nothing."""
  from advoc_amd import conv
  hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_H3_PATCH_MIN_WGS=1, ADVOC_WGRAD_H3_MIN_M=1)
  dev = torch.device('cuda')
  g = torch.Generator().manual_seed(91)
  if producer == 'thin':          # layer_1 -> layer_2: two 1-channel sources, thin matrix kernel
    c0 = torch.randn(2, 64, 129, 1, generator=g).to(dev)
    c1 = torch.randn(2, 64, 129, 1, generator=g).to(dev)
    cin, cmid, cout, s2, y1s, y2s, act1 = 2, 64, 128, (2, 2), (2, 32, 64, 64), (2, 16, 32, 128), conv.ACT_NONE
    want1, want2 = 'thin_k_gemm_kernel', 'patch_gemm_h3_kernel<4, 1>'
  elif producer == 'patch_s2':    # layer_2 -> layer_3: the producer's forward on the stride-2 patch kernel (128 columns)
    c0 = torch.randn(2, 64, 128, 64, generator=g).to(dev)
    c1 = None
    cin, cmid, cout, s2, y1s, y2s, act1 = 64, 128, 256, (2, 2), (2, 32, 64, 128), (2, 16, 32, 256), conv.ACT_LRELU
    want1, want2 = 'patch_gemm_h3_kernel<3, 0>', 'patch_gemm_h3_kernel<4, 1>'
  else:                           # layer_3 -> layer_4: 256 columns forward, the reader a 4x4 stride-1 layer
    c0 = torch.randn(2, 64, 64, 128, generator=g).to(dev)
    c1 = None
    cin, cmid, cout, s2, y1s, y2s, act1 = 128, 256, 256, (1, 1), (2, 32, 32, 256), (2, 31, 31, 256), conv.ACT_LRELU
    want1, want2 = 'patch_gemm_h3_kernel<2, 0>', 'patch_gemm_h3_kernel<1, 1>'
  w1 = (torch.randn(4, 4, cin, cmid, generator=g) * (0.2 if cin == 2 else 0.05)).to(dev)
  b1 = (torch.randn(cmid, generator=g) * 0.1).to(dev)
  w2 = (torch.randn(4, 4, cmid, cout, generator=g) * 0.05).to(dev)
  b2 = (torch.randn(cout, generator=g) * 0.1).to(dev)
  dy2 = torch.randn(*y2s, generator=g).to(dev)

  def make(exclusive):
    y1 = torch.full(y1s, float('nan'), device=dev)
    x0, x1 = c0.clone(), (c1.clone() if c1 is not None else None)
    L1 = conv.Layer(conv.CONV, x0, y1, w1, b1, x1=x1, stride=(2, 2), pad=(1, 1), in_act=act1,
                    w_amax=w1.abs().max().reshape(1).view(torch.int32))
    L2 = conv.Layer(conv.CONV, y1, torch.empty(*y2s, device=dev), w2, b2, stride=s2, pad=(1, 1),
                    in_act=conv.ACT_LRELU)
    for L in (L1, L2):
      L.delayed_scale, L.reuse_images = True, True
    assert want1 in L1.kernel_name(0) and L2.kernel_name(1) == want2, (L1.kernel_name(0), L2.kernel_name(1))
    L1.add_image_consumer(L2, 0, exclusive=exclusive)
    t = dict(x0=x0, x1=x1, y1=y1, dx=torch.empty_like(y1), dw=torch.zeros_like(w2), db=torch.zeros_like(b2))
    return L1, L2, t
  A, R = make(True), make(False)
  for step, scale in enumerate((1.0, 0.9, 1000.0, 1.0)):
    for (L1, L2, t), only in ((A, True), (R, False)):
      t['x0'].copy_(c0 * scale)
      if c1 is not None:
        t['x1'].copy_(c1 * scale)
      t['y1'].fill_(float('nan'))
      L1.forward()
      if only:
        assert L2._x_final and L2._x_gates, step                    # from the first step on
        assert bool(torch.isnan(t['y1']).all()), step               # the fp32 tensor is NOT written
      else:
        assert not L2._x_gates and not bool(torch.isnan(t['y1']).any()), step
      L2.forward()
      t['dw'].zero_()
      L2.backward_data(dy2, t['dx'], db=t['db'], db_accumulate=False)
      L2.backward_weight(dy2, t['dw'], t['db'])
    (_, L2a, ta), (_, L2r, tr) = A, R
    assert rel(L2a.y, L2r.y) < 2e-6, (step, rel(L2a.y, L2r.y))
    assert rel(ta['dx'], tr['dx']) < 2e-6, (step, rel(ta['dx'], tr['dx']))
    assert rel(ta['dw'], tr['dw']) < 2e-6, (step, rel(ta['dw'], tr['dw']))
    assert rel(ta['db'], tr['db'].double().cpu()) < 2e-6, (step, rel(ta['db'], tr['db'].double().cpu()))
    # the image's header: word 0 = the largest magnitude of lrelu(y) written, word 1 = 2^-s with the bound at [2^14, 2^15)
    ha = L2a.image_headers()[0].cpu().view(torch.float32)
    amax = float(torch.nn.functional.leaky_relu(tr['y1'], 0.2).abs().max())
    assert abs(float(ha[0]) - amax) <= 2e-6 * amax, (step, float(ha[0]), amax)
    assert 2.0 <= amax / float(ha[1]) < 65504.0, (step, amax / float(ha[1]))
  assert int(L2a.image_headers()[0].cpu()[5]) == 0      # nothing was ever refitted
  # (r6) an input that exists as its image only stays CURRENT until the next forward: a second backward_weight after
  # the same forward (gradient accumulation, a profiler's re-run) reads the image again and reproduces the gradient -- it
  # used to rebuild the image from the never-written fp32 tensor (NaN poison here) and overwrite the one valid copy ...
  _, L2a, ta = A
  again = torch.zeros_like(ta['dw'])
  L2a.backward_data(dy2, ta['dx'])
  L2a.backward_weight(dy2, again)
  assert torch.isfinite(again).all() and rel(again, ta['dw']) < 1e-6, rel(again, ta['dw'])
  L2a.backward_weight(dy2, again)
  assert torch.isfinite(again).all() and rel(again, ta['dw']) < 1e-6
  # ... and a call that could only rebuild it is refused, in Python and by the library itself
  from advoc_amd import _lib
  L2a._x_current = False
  with pytest.raises(_lib.AdvocHipError, match='image only'):
    L2a.backward_weight(dy2, again)
  import ctypes
  L2a.struct.img_flags = 256            # ADVOC_IMG_X_GATES without ADVOC_IMG_X_CURRENT
  rc = _lib.load().advoc_conv_backward_weight(ctypes.byref(L2a.struct), _lib.ptr(dy2), _lib.ptr(again), None, 0, _lib.stream())
  L2a.struct.img_flags = 0
  assert rc != 0


@gpu
def test_output_gradient_roles_keep_separate_magnitude_histories(hip, hipenv):
  """Layer.set_dy_role: one layer object that sees gradients of two losses per step (the discriminator's fake pass with
  batch norm: D-loss gradients in the D step, ~1000 x larger G-loss gradients in the G step) keeps one header per role,
  so neither sequence ever leaves its own one-pass window (no refits after the first image of each role)."""
  from advoc_amd import conv
  hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_WGRAD_H3_MIN_M=1)
  dev = torch.device('cuda')
  g = torch.Generator().manual_seed(23)
  x = torch.randn(2, 16, 33, 128, generator=g).to(dev)
  w = (torch.randn(4, 4, 128, 256, generator=g) * 0.05).to(dev)
  dy = torch.randn(2, 8, 17, 256, generator=g).to(dev)
  y = torch.empty(2, 8, 17, 256, device=dev)
  L = conv.Layer(conv.CONV, x, y, w, None, stride=(2, 2), pad=(1, 1), in_act=conv.ACT_LRELU)
  L.delayed_scale, L.reuse_images = True, True
  E = conv.Layer(conv.CONV, x, torch.empty_like(y), w, None, stride=(2, 2), pad=(1, 1), in_act=conv.ACT_LRELU)
  for it in range(3):
    for role, scale in (('d', 1e-3), ('g', 1.0)):
      L.set_dy_role(role)
      dx, dxe = torch.zeros_like(x), torch.zeros_like(x)
      L.backward_data(dy * scale, dx)
      E.backward_data(dy * scale, dxe)
      assert rel(dx, dxe) < 2e-6
  hdrs = L.image_headers()
  assert len(hdrs) == 3                              # x, dy role 'd', dy role 'g'
  assert sum(int(h.cpu()[5]) for h in hdrs) == 0     # never out of window
  assert all(int(h.cpu()[2]) != 0 for h in hdrs[1:])   # both roles ran the one-pass form


@gpu
@pytest.mark.parametrize('case', [c for c in H3 if c[0] in ('h3_enc', 'h3_dec_skip')] +
                         [('bias_n192', 0, (3, 16, 33), 64, 0, 192, 0, (2, 2), None, 1, True, 0)] +
                         [c[0] for c in PATCH if c[0][0] in ('p3_dec_skip', 'p3_enc_bwd_odd', 'p3_rem_dec')],
                         ids=lambda c: c[0])
def test_bias_gradient_rides_in_the_output_gradient_image_pass(hip, case, hipenv):
  """backward_data(..., db=...) on the image kernels: the per-channel sums of dy (times the forward dropout mask, logical
  columns only) are taken by the pass that writes dy's fp16 pair image, the following backward_weight skips its bias
  kernel; same numbers as advoc_conv_backward_bias and the float64 oracle, with and without accumulation, in exact and
  in delayed-scale image mode.  192 output channels (256 % 24 != 0): not fusable, the bias kernel runs as before."""
  from advoc_amd import conv
  hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_H3_PATCH_MIN_WGS=1, ADVOC_WGRAD_H3_MIN_M=1)
  c = build_case(case)
  dev = torch.device('cuda')
  _, _, _, _, db_o = oracle_layer(c['kind'], c['x0'], c['x1'], c['in_w'], c['w'], c['b'], c['stride'], c['pad'], c['act'],
                                  c['mask'], c['keep'], c['out_w'], c['dy'])
  x0 = c['x0'].to(dev)
  x1 = c['x1'].to(dev) if c['x1'] is not None else None
  w, b, dy = c['w'].to(dev), c['b'].to(dev), c['dy'].to(dev)
  mask = c['mask'].to(dev) if c['mask'] is not None else None
  cout = w.shape[3] if c['kind'] == 0 else w.shape[2]
  y = torch.empty(x0.shape[0], c['oh'], c['out_w'], cout, device=dev)
  L = conv.Layer(c['kind'], x0, y, w, b, x1=x1, in_w=c['in_w'], stride=c['stride'], pad=c['pad'], in_act=c['act'],
                 drop_mask=mask, drop_scale=1 / c['keep'] if mask is not None else 0.)
  L.reuse_images = True
  L.delayed_scale = True
  assert 'h3' in L.kernel_name(1)
  assert L._bias_fusable == (256 % (cout // 8) == 0)
  L.forward()
  dx0 = torch.zeros_like(x0)
  dx1 = torch.zeros_like(x1) if x1 is not None else None
  for rep in range(3):                      # first call: exact two-pass image; then the delayed one-pass form
    db = torch.full_like(b, 5.0)
    dw = torch.zeros_like(w)
    L.backward_data(dy, dx0, dx1, db=db, db_accumulate=(rep == 2))
    if L._bias_fusable:
      assert L._db_done_for == (dy.data_ptr(), db.data_ptr())
      assert rel(db - (5.0 if rep == 2 else 0.0), db_o) < TOL, (rep, rel(db, db_o))     # already there
    L.backward_weight(dy, dw, db, accumulate=(rep == 2))
    assert rel(db - (5.0 if rep == 2 else 0.0), db_o) < TOL, (rep, rel(db, db_o))


@gpu
def test_segmented_amax_of_an_arena(hip):
  """advoc_segmented_amax_f32: max |w| of many tensors of one flat buffer in one launch, as float bits; segments that are
  empty, all zero, one element long, not 16-byte aligned, or longer than one grid sweep."""
  import ctypes
  from advoc_amd import _lib
  dev = torch.device('cuda')
  g = torch.Generator().manual_seed(5)
  flat = torch.randn(3_000_000, generator=g).to(dev)
  segs = [(0, 16), (16, 1), (17, 4099), (4116, 0), (4120, 64), (8192, 2_500_000), (2_600_000, 4 * 4 * 256 * 64)]
  flat[4120:4184] = 0.
  flat[9000] = -77.5
  off = torch.tensor([s[0] for s in segs], dtype=torch.int64, device=dev)
  size = torch.tensor([s[1] for s in segs], dtype=torch.int64, device=dev)
  out = torch.full((len(segs),), 123, dtype=torch.int32, device=dev)
  _lib.check(_lib.load().advoc_segmented_amax_f32(_lib.ptr(flat), _lib.ptr(off), _lib.ptr(size), len(segs), _lib.ptr(out),
                                                  _lib.stream()), 'advoc_segmented_amax_f32')
  got = out.view(torch.float32).cpu()
  for i, (o, n) in enumerate(segs):
    want = float(flat[o:o + n].abs().max()) if n else 0.0
    assert float(got[i]) == want, (i, float(got[i]), want)
  assert float(got[5]) == 77.5


@gpu
@pytest.mark.parametrize('case', [c for c in H3 if c[0] in ('h3_enc', 'h3_dec_skip')] +
                         [c[0] for c in PATCH if c[0][0] in ('p3_dec_skip', 'p3_enc_bwd_odd')], ids=lambda c: c[0])
def test_weight_magnitude_from_the_arena_pass(hip, case, hipenv):
  """advoc_conv_layer.w_amax: with the largest |w| handed in (advoc_segmented_amax_f32) the image kernels skip their own
  magnitude pass over the kernel and give bit-identical results, forward and backward-data, also after the kernel
  changed (refreshed w_amax); so do the persistent weight images of advoc_weight_images_f32 (w_img / w_img_hdr)."""
  import ctypes
  from advoc_amd import conv, _lib
  hipenv(ADVOC_H3_MIN_TILES=1, ADVOC_H3_PATCH_MIN_WGS=1)
  c = build_case(case)
  dev = torch.device('cuda')
  x0 = c['x0'].to(dev)
  x1 = c['x1'].to(dev) if c['x1'] is not None else None
  w, dy = c['w'].to(dev).clone(), c['dy'].to(dev)
  cout = w.shape[3] if c['kind'] == 0 else w.shape[2]
  y = torch.empty(x0.shape[0], c['oh'], c['out_w'], cout, device=dev)
  amax = torch.zeros(1, dtype=torch.int32, device=dev)
  off = torch.zeros(1, dtype=torch.int64, device=dev)
  size = torch.full((1,), w.numel(), dtype=torch.int64, device=dev)

  def refresh():
    _lib.check(_lib.load().advoc_segmented_amax_f32(_lib.ptr(w), _lib.ptr(off), _lib.ptr(size), 1, _lib.ptr(amax),
                                                    _lib.stream()), 'advoc_segmented_amax_f32')

  def run(w_amax):
    L = conv.Layer(c['kind'], x0, y, w, None, x1=x1, in_w=c['in_w'], stride=c['stride'], pad=c['pad'], in_act=c['act'],
                   w_amax=w_amax)
    assert 'h3' in L.kernel_name(0) and 'h3' in L.kernel_name(1)
    y.zero_()
    L.forward()
    dx0 = torch.zeros_like(x0)
    dx1 = torch.zeros_like(x1) if x1 is not None else None
    L.backward_data(dy, dx0, dx1)
    return y.clone(), dx0.clone()

  def run_with_persistent_images():
    """advoc_weight_images_f32: both weight images of the layer from one launch, handed in as w_img / w_img_hdr"""
    L = conv.Layer(c['kind'], x0, y, w, None, x1=x1, in_w=c['in_w'], stride=c['stride'], pad=c['pad'], in_act=c['act'],
                   w_amax=amax)
    descs = [L.weight_image_desc(d) for d in (0, 1)]
    assert all(d is not None for d in descs) and descs[0][3] != descs[1][3]        # one [tap][k][n], one [tap][n][k]
    offs = [0, descs[0][4]]
    pool = torch.empty(descs[0][4] + descs[1][4], dtype=torch.uint8, device=dev)
    hdrs = torch.zeros(8, dtype=torch.int32, device=dev)
    table = torch.tensor([[0, d[0], d[1], d[2], d[3], 0, offs[i], i] for i, d in enumerate(descs)], dtype=torch.int64,
                         device=dev)
    _lib.check(_lib.load().advoc_weight_images_f32(_lib.ptr(w), _lib.ptr(amax), _lib.ptr(table), 2, _lib.ptr(pool),
                                                   _lib.ptr(hdrs), _lib.stream()), 'advoc_weight_images_f32')
    for d in (0, 1):
      L.set_weight_image(d, pool.data_ptr() + offs[d], hdrs.data_ptr() + 16 * d)
    # (r5) advoc_weight_images_l1_f32: the same images, and behind {max |w|, 2^-s} in 32-word headers the per-tap maxima over
    # the image's rows n of sum_k |w[tap][n][k]| -- the factors of the a-priori bounds of the emitting launches
    pool2 = torch.empty_like(pool)
    hdrs2 = torch.zeros(64, dtype=torch.int32, device=dev)
    _lib.check(_lib.load().advoc_weight_images_l1_f32(_lib.ptr(w), _lib.ptr(amax), _lib.ptr(table), 2, _lib.ptr(pool2),
                                                      _lib.ptr(hdrs2), _lib.stream()), 'advoc_weight_images_l1_f32')
    assert torch.equal(pool, pool2)
    h2 = hdrs2.cpu()
    for i, d in enumerate(descs):
      taps, n_total, ktot, b_kn = d[:4]
      assert int(h2[32 * i + 2]) == taps and int(h2[32 * i + 3]) == ktot and int(h2[32 * i]) == int(amax) and \
          int(h2[32 * i + 1]) == int(hdrs.cpu()[4 * i + 1])
      wt = w.detach().double().cpu().reshape(taps, -1)
      wt = wt.reshape(taps, ktot, n_total) if b_kn else wt.reshape(taps, n_total, ktot).transpose(1, 2)
      want = wt.abs().sum(dim=1).max(dim=1).values                   # [tap]: max over n of the sum over k
      got = h2[32 * i + 4:32 * i + 4 + taps].view(torch.float32).double()
      assert bool(((got >= want) & (got <= want * (1 + 3e-5) + 1e-11)).all()), (got, want)    # an upper bound, and a tight one
    y.zero_()
    L.forward()
    dx0 = torch.zeros_like(x0)
    dx1 = torch.zeros_like(x1) if x1 is not None else None
    L.backward_data(dy, dx0, dx1)
    return y.clone(), dx0.clone()

  for rep in range(2):
    refresh()
    assert float(amax.view(torch.float32)) == float(w.abs().max())
    a, b, pi = run(None), run(amax), run_with_persistent_images()
    for u, v, z in zip(a, b, pi):
      assert torch.equal(u, v)
      assert torch.equal(u, z)
    w.mul_(37.0)                          # "optimizer step": another power of two
  with pytest.raises(_lib.AdvocHipError):
    conv.Layer(c['kind'], x0, y, w, None, x1=x1, in_w=c['in_w'], stride=c['stride'], pad=c['pad'], in_act=c['act'],
               w_amax=torch.zeros(2, dtype=torch.int32, device=dev))
