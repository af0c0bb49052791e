"""Which kernel each layer of AdVoc-full at B=64 (BASELINE configs[2]) launches, decided on the host: runs without a
GPU (advoc_conv_kernel_name only walks the dispatch rules; pointers are never dereferenced, the CU count falls back to
256).  Guards the rules of conv.hip / igemm_h3.hip / igemm_patch.hip / wgrad_h3.hip against silent regressions -- the
GPU suite then checks the same names on real tensors (tests/test_hip_fullsize.py)."""
import ctypes

import pytest

from advoc_amd import _lib

BF = 64
FAKE = 0x10000000      # never dereferenced


def layer(kind, n, h, w, c0, c1, cout, stride, trim=0):
  L = _lib.ConvLayer()
  L.kind, L.kh, L.kw, L.sh, L.sw, L.pad_t, L.pad_l, L.in_act = kind, 4, 4, stride[0], stride[1], 1, 1, 1
  L.x0 = _lib.Tensor4(FAKE, n, h, w, c0, w + trim)
  if c1:
    L.x1 = _lib.Tensor4(FAKE + (1 << 20), n, h, w, c1, w)
  if kind == 0:
    oh = (h + 2 - 4) // stride[0] + 1
    ow = (w + 2 - 4) // stride[1] + 1
  else:
    oh, ow = 2 * h, 2 * w
  L.y = _lib.Tensor4(FAKE + (2 << 20), n, oh, ow, cout, ow)
  L.w = FAKE + (3 << 20)
  L.workspace = FAKE + (4 << 20)
  L.workspace_bytes = 1 << 40
  # persistent operand images as conv.Layer allocates them where the image-based weight gradient applies
  L.x_img, L.x_hdr, L.dy_img, L.dy_hdr = FAKE + (5 << 20), FAKE + (6 << 20), FAKE + (7 << 20), FAKE + (8 << 20)
  return L


def names(L):
  lib = _lib.load()
  out = []
  for d in range(3):
    buf = ctypes.create_string_buffer(128)
    assert lib.advoc_conv_kernel_name(ctypes.byref(L), d, buf, 128) == 0
    out.append(buf.value.decode())
  return out


P4F, P4B = 'patch_gemm_h3_kernel<4, 0>', 'patch_gemm_h3_kernel<4, 1>'
P1F, P1B = 'patch_gemm_h3_kernel<1, 0>', 'patch_gemm_h3_kernel<1, 1>'
P2F, P2B = 'patch_gemm_h3_kernel<2, 0>', 'patch_gemm_h3_kernel<2, 1>'
P3F = 'patch_gemm_h3_kernel<3, 0>'
W256, W128 = 'wgrad_h3_256_kernel', 'wgrad_h3_kernel'
# (r6) grid rows under 32 points: the `_flat` instances under their own names (as rocprofv3 lists them)
W256F, W128F = 'wgrad_h3_256_flat_kernel', 'wgrad_h3_flat_kernel'

# (name, layer, forward, backward-data, backward-weight); None = not asserted (edge / tiny layers on the r1 kernels)
FULL = [
    ('encoder_2', layer(0, BF, 128, 257, 64, 0, 128, (2, 2)), P3F, P4B, W128),
    ('encoder_3', layer(0, BF, 64, 129, 128, 0, 256, (2, 2)), P2F, P4B, W256),
    ('encoder_4', layer(0, BF, 32, 65, 256, 0, 512, (2, 2)), P2F, P4B, W256),
    # (r4: under one round of tiles the tile follows the launch's shape, igemm_h3.hip pick_tile -- encoder_5 forward, 68
    # 256 x 256 tiles, on that tile cut into three K slices; its four-phase backward-data and decoder_5 forward, 272 such
    # tiles on 256 CUs, on 128 x 128 tiles; decoder_6 backward-data, 144 128 x 128 tiles, on those cut into three K slices)
    ('encoder_5', layer(0, BF, 16, 33, 512, 0, 512, (2, 2)), 'gather_gemm_h3_kernel<2, 4, 2, 4>',
     'gather_gemm_h3_kernel<2, 2, 2, 2>', W256F),
    ('decoder_6', layer(1, BF, 4, 9, 512, 512, 512, (2, 2), trim=1), 'gather_gemm_h3_kernel<2, 1, 3, 2>',
     'gather_gemm_h3_kernel<2, 2, 2, 2>', None),
    ('encoder_7', layer(0, BF, 4, 9, 512, 0, 512, (2, 2)), 'gather_gemm_h3_kernel<2, 1, 3, 2>',
     'gather_gemm_h3_kernel<2, 1, 3, 2>', None),
    ('decoder_5', layer(1, BF, 8, 17, 512, 512, 512, (2, 2), trim=1), 'gather_gemm_h3_kernel<2, 2, 2, 2>',
     'gather_gemm_h3_kernel<2, 2, 2, 2>', W256F),
    ('decoder_4', layer(1, BF, 16, 33, 512, 512, 256, (2, 2), trim=1), P4F, P2B, W256),
    ('decoder_3', layer(1, BF, 32, 65, 256, 256, 128, (2, 2), trim=1), P4F, P2B, W256),
    ('decoder_2', layer(1, BF, 64, 129, 128, 128, 64, (2, 2), trim=1), P4F, P2B, W256),
    ('layer_2', layer(0, 2 * BF, 128, 256, 64, 0, 128, (2, 2)), P3F, P4B, W128),
    ('layer_3', layer(0, 2 * BF, 64, 128, 128, 0, 256, (2, 2)), P2F, P4B, W256),
    ('layer_4', layer(0, 2 * BF, 32, 64, 256, 0, 512, (1, 1)), P1F, P1B, W256),
]


@pytest.mark.parametrize('name,L,fwd,bwd,wgt', FULL, ids=[c[0] for c in FULL])
def test_full_model_dispatch(name, L, fwd, bwd, wgt):
  got = names(L)
  for d, want in enumerate((fwd, bwd, wgt)):
    if want is not None:
      assert got[d] == want, (name, d, got[d], want)


BS = 32
PT = 'gather_gemm_h3_kernel<2, 1, 3, 2>'
# AdVoc-small at B=32 (BASELINE configs[1], advoc_model_small.py:14-15: ngf = ndf = 32): every layer between the 1-channel
# edges runs on the image kernels -- the 32-column launches (encoder_2 / layer_2 backward-data, decoder_2 forward) on the
# 64-column four-phase patch instance without the column block that does not exist (r4, second half; the 128 x 64 per-tap
# tile with its upper half masked before that, the r1 fp32 kernel before that)
SMALL = [
    ('encoder_2', layer(0, BS, 128, 257, 32, 0, 64, (2, 2)), PT, P4B, W128),
    ('encoder_3', layer(0, BS, 64, 129, 64, 0, 128, (2, 2)), P3F, P4B, W128),
    ('decoder_3', layer(1, BS, 32, 65, 128, 128, 64, (2, 2), trim=1), P4F, P2B, W128),
    ('decoder_2', layer(1, BS, 64, 129, 64, 64, 32, (2, 2), trim=1), P4F, 'patch_gemm_h3_kernel<3, 1>', W128),
    ('layer_2', layer(0, 2 * BS, 128, 256, 32, 0, 64, (2, 2)), PT, P4B, W128),
    ('layer_3', layer(0, 2 * BS, 64, 128, 64, 0, 128, (2, 2)), P3F, P4B, W128),
    ('layer_4', layer(0, 2 * BS, 32, 64, 128, 0, 256, (1, 1)), P1F, 'patch_gemm_h3_kernel<5, 1>', W256),
]


@pytest.mark.parametrize('name,L,fwd,bwd,wgt', SMALL, ids=[c[0] for c in SMALL])
def test_small_model_dispatch(name, L, fwd, bwd, wgt):
  got = names(L)
  for d, want in enumerate((fwd, bwd, wgt)):
    if want is not None:
      assert got[d] == want, (name, d, got[d], want)
  assert not any(n.startswith('gather_gemm_kernel') or 'wgrad_mfma' in n for n in got), got


def test_edge_layers_keep_their_direct_kernels():
  """1-channel inputs / outputs never take the image kernels (channel counts are not multiples of 32)."""
  enc1 = layer(0, BF, 256, 513, 1, 0, 64, (2, 2))
  dec1 = layer(1, BF, 128, 257, 64, 64, 1, (2, 2), trim=1)
  for L in (enc1, dec1):
    for n in names(L):
      assert 'h3' not in n, n
