// C-ABI entry points of the convolution stack: validate an advoc_conv_layer, translate each
// direction into a gather-GEMM / weight-gradient problem and pick the kernel: the fp32-MFMA
// implicit GEMM (igemm.hip, wgrad.hip) for the wide layers, the direct HBM-bound kernels
// (edge.hip, wgrad.hip thin path) for the layers with 1-2 input channels or 1 output channel.
//
// Reference graph being replaced: models/advoc/advoc_model.py:25-69 (layer builders),
// :89-158 (generator), :184-202 (discriminator) and their TF gradients.
#include "conv_internal.h"
#include "tuning.h"
#include <string.h>

#include "x6.h"

namespace advoc {

static_assert(kColsumBytes == ADVOC_WGRAD_TABLE_BYTES, "advoc_conv_layer.wgrad_table size");

static inline int pack_tap(int dy, int dx, int wtap) {
  return (dy & 0xff) | ((dx & 0xff) << 8) | (wtap << 16);
}

int validate_layer(const advoc_conv_layer* L) {
  if (!L) return ADVOC_ERR_NULL;
  if (!L->x0.p || !L->y.p || !L->w) return ADVOC_ERR_NULL;
  if (L->kind != ADVOC_CONV && L->kind != ADVOC_DECONV) return ADVOC_ERR_UNSUPPORTED;
  const advoc_tensor4 &x0 = L->x0, &x1 = L->x1, &y = L->y;
  if (x0.n <= 0 || x0.h <= 0 || x0.w <= 0 || x0.c <= 0 || x0.w_pitch < x0.w) return ADVOC_ERR_BAD_SHAPE;
  if (y.n != x0.n || y.h <= 0 || y.w <= 0 || y.c <= 0 || y.w_pitch < y.w) return ADVOC_ERR_BAD_SHAPE;
  if (x1.p) {
    if (x1.n != x0.n || x1.h != x0.h || x1.w != x0.w || x1.c <= 0 || x1.w_pitch < x1.w)
      return ADVOC_ERR_BAD_SHAPE;
  }
  // dense tap lists hold kMaxTaps entries; a transposed conv is run phase by phase (<= ceil(k/s)^2 taps
  // each), so its forward direction also takes 5x5 kernels (MelspecGAN, models/melspecgan/conv2d.py:17-53)
  const int max_taps = L->kind == ADVOC_DECONV ? 25 : kMaxTaps;
  if (L->kh < 1 || L->kw < 1 || L->kh * L->kw > max_taps || L->sh < 1 || L->sw < 1) return ADVOC_ERR_UNSUPPORTED;
  if (L->pad_t < 0 || L->pad_l < 0 || L->pad_t >= L->kh || L->pad_l >= L->kw) return ADVOC_ERR_UNSUPPORTED;
  if (L->in_act < ADVOC_ACT_NONE || L->in_act > ADVOC_ACT_RELU) return ADVOC_ERR_UNSUPPORTED;
  if ((L->in_scale == nullptr) != (L->in_shift == nullptr)) return ADVOC_ERR_NULL;
  if (L->kind == ADVOC_CONV) {
    // every output pixel's window must start inside the padded input
    if ((int64_t)(y.h - 1) * L->sh - L->pad_t >= x0.h || (int64_t)(y.w - 1) * L->sw - L->pad_l >= x0.w)
      return ADVOC_ERR_BAD_SHAPE;
  } else {
    // strides (2,2), and (1,2) for the layers the reference builds once the time axis has
    // shrunk to 1 (advoc_model.py:109-116,139-142)
    if (L->kh < 4 || L->kh > 5 || L->kw != L->kh || L->sh < 1 || L->sh > 2 || L->sw < 1 || L->sw > 2 ||
        L->pad_t != 1 || L->pad_l != 1)
      return ADVOC_ERR_UNSUPPORTED;
    if (y.h != L->sh * x0.h || y.w > L->sw * x0.w || y.w < L->sw * x0.w - 1) return ADVOC_ERR_BAD_SHAPE;
  }
  return ADVOC_OK;
}

namespace {

int cin_of(const advoc_conv_layer* L) { return L->x0.c + (L->x1.p ? L->x1.c : 0); }

// taps of one sub-pixel phase of a stride-s transposed gather (s = 1 or 2):
//   out index o = s g + par reads in index g + d for every k with s | (par + pad - k),
//   d = (par + pad - k) / s
int phase_taps(int par, int pad, int ksize, int s, int* ks, int* ds) {
  int n = 0;
  for (int k = 0; k < ksize; ++k) {
    const int t = par + pad - k;
    if (s == 1 || (t & 1) == 0) {
      ks[n] = k;
      ds[n] = t / s;   // exact (t even when s == 2), also for negatives
      ++n;
    }
  }
  return n;
}

void dense_taps(GatherGemmParams& p, const advoc_conv_layer* L, bool flipped) {
  p.nphase = 1;
  p.ntaps = L->kh * L->kw;
  for (int ky = 0; ky < L->kh; ++ky)
    for (int kx = 0; kx < L->kw; ++kx)
      p.tap[0][ky * L->kw + kx] = flipped ? pack_tap(L->pad_t - ky, L->pad_l - kx, ky * L->kw + kx)
                                          : pack_tap(ky - L->pad_t, kx - L->pad_l, ky * L->kw + kx);
  p.osy = p.osx = 1;
}

int subpixel_taps(GatherGemmParams& p, const advoc_conv_layer* L) {
  if (L->sh > 2 || L->sw > 2) return ADVOC_ERR_UNSUPPORTED;
  p.sy = p.sx = 1;
  p.osy = L->sh;
  p.osx = L->sw;
  p.nphase = L->sh * L->sw;
  int counts[kMaxPhases] = {0, 0, 0, 0};
  int most = 0;
  for (int py = 0; py < L->sh; ++py)
    for (int px = 0; px < L->sw; ++px) {
      int kys[kMaxTaps], dys[kMaxTaps], kxs[kMaxTaps], dxs[kMaxTaps];
      const int ny = phase_taps(py, L->pad_t, L->kh, L->sh, kys, dys);
      const int nx = phase_taps(px, L->pad_l, L->kw, L->sw, kxs, dxs);
      if (ny * nx > kMaxTaps) return ADVOC_ERR_UNSUPPORTED;
      const int ph = py * L->sw + px;
      counts[ph] = ny * nx;
      most = ny * nx > most ? ny * nx : most;
      for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b)
          p.tap[ph][a * nx + b] = pack_tap(dys[a], dxs[b], kys[a] * L->kw + kxs[b]);
      p.ooy[ph] = py;
      p.oox[ph] = px;
    }
  p.ntaps = most;
  // Odd kernels give the phases different tap counts (5x5, stride 2: 9 / 6 / 6 / 4).  The kernels walk
  // ONE count, so the short phases are padded with taps that read 128 rows AND columns before the
  // grid point -- outside any input this is accepted for, i.e. they contribute exactly zero.
  for (int ph = 0; ph < p.nphase; ++ph)
    for (int t = counts[ph]; t < most; ++t) {
      if (p.in_h > 128 && p.in_w > 128) return ADVOC_ERR_UNSUPPORTED;
      p.tap[ph][t] = pack_tap(-128, -128, 0);
    }
  return ADVOC_OK;
}

// ---------------------------------------------------------------------------------------------
// forward:   K = cin, N = cout
// ---------------------------------------------------------------------------------------------
int build_forward(const advoc_conv_layer* L, GatherGemmParams& p, bool& b_kn) {
  p = GatherGemmParams{};
  p.a0 = L->x0.p; p.c0 = L->x0.c; p.a0_pitch = L->x0.w_pitch;
  p.a1 = L->x1.p; p.c1 = L->x1.p ? L->x1.c : 0; p.a1_pitch = L->x1.p ? L->x1.w_pitch : 0;
  p.a_h = L->x0.h; p.in_h = L->x0.h; p.in_w = L->x0.w;
  p.in_scale = L->in_scale; p.in_shift = L->in_shift; p.in_act = L->in_act;
  p.a_mask = L->in_mask; p.a_mask_scale = L->in_mask_scale;
  p.batch = L->x0.n;
  p.w = L->w;
  p.n_total = p.n_split = L->y.c;
  p.d[0].p = L->y.p; p.d[0].pitch = L->y.w_pitch; p.d[0].c = L->y.c;
  p.out_h = L->y.h; p.out_w = L->y.w;
  p.bias = L->b;
  p.y_mask = L->drop_mask; p.y_mask_scale = L->drop_scale;
  if (L->kind == ADVOC_CONV) {
    p.gh = L->y.h; p.gw = L->y.w;
    p.sy = L->sh; p.sx = L->sw;
    dense_taps(p, L, false);
    b_kn = true;    // kernel [kh,kw,ci,co]: K rows, N contiguous
    return ADVOC_OK;
  }
  // transposed conv: sh x sw dense sub-pixel phases over the INPUT grid
  p.gh = L->x0.h; p.gw = L->x0.w;
  b_kn = false;     // kernel [kh,kw,co,ci]: N rows, K contiguous
  return subpixel_taps(p, L);
}

// ---------------------------------------------------------------------------------------------
// backward data:   K = cout, N = cin
// ---------------------------------------------------------------------------------------------
int build_backward_data(const advoc_conv_layer* L, const float* dy, float* dx0, float* dx1, int accum0,
                        int accum1, GatherGemmParams& p, bool& b_kn) {
  p = GatherGemmParams{};
  p.a0 = dy; p.c0 = L->y.c; p.a0_pitch = L->y.w_pitch;
  p.a_h = L->y.h; p.in_h = L->y.h; p.in_w = L->y.w;
  p.in_act = ADVOC_ACT_NONE;
  p.a_mask = L->drop_mask; p.a_mask_scale = L->drop_scale;
  p.batch = L->x0.n;
  p.w = L->w;
  const int c0 = L->x0.c, c1 = L->x1.p ? L->x1.c : 0;
  p.n_total = c0 + c1;
  p.n_split = c0;
  p.d[0].p = dx0; p.d[0].xpre = L->x0.p; p.d[0].pitch = L->x0.w_pitch; p.d[0].c = c0; p.d[0].accum = accum0;
  p.d[1].p = c1 ? dx1 : nullptr; p.d[1].xpre = L->x1.p; p.d[1].pitch = c1 ? L->x1.w_pitch : 0;
  p.d[1].c = c1; p.d[1].accum = accum1;
  p.d[0].gscale = L->in_scale; p.d[0].gshift = L->in_shift;
  p.d[0].gmask = L->in_mask; p.d[0].gmask_scale = L->in_mask_scale;
  p.d[1].gscale = L->in_scale ? L->in_scale + c0 : nullptr;
  p.d[1].gshift = L->in_shift ? L->in_shift + c0 : nullptr;
  p.out_h = L->x0.h; p.out_w = L->x0.w;
  p.grad_act = L->in_act;
  if (L->kh * L->kw > kMaxTaps) return ADVOC_ERR_UNSUPPORTED;   // 5x5 transposed convs: forward (inference) only
  if (L->kind == ADVOC_DECONV) {
    // dIn[iy,ix,ci] = sum dOut[2 iy - pad + ky, 2 ix - pad + kx, co] * w[ky,kx,co,ci]
    p.gh = L->x0.h; p.gw = L->x0.w;
    p.sy = L->sh; p.sx = L->sw;
    dense_taps(p, L, false);
    b_kn = true;    // [tap][co = K][ci = N]
    return ADVOC_OK;
  }
  b_kn = false;     // [tap][ci = N][co = K]
  if (L->sh == 1 && L->sw == 1) {
    // dX[iy,ix] = sum dY[iy + pad - ky, ix + pad - kx] * w[ky,kx]
    p.gh = L->x0.h; p.gw = L->x0.w;
    p.sy = p.sx = 1;
    dense_taps(p, L, true);
    return ADVOC_OK;
  }
  p.gh = (L->x0.h + L->sh - 1) / L->sh; p.gw = (L->x0.w + L->sw - 1) / L->sw;
  return subpixel_taps(p, L);
}

// <= 2 output columns, wide K: S[pixel][tap * N + n] = sum_k A[pixel, k] w[tap][n][k] for every
// input pixel ONCE (pointwise MFMA GEMM, taps become GEMM columns), then a gather-sum over taps.
// The direct kernel re-reads each input pixel once per tap that touches it (4-16x).
bool two_stage_ok(const GatherGemmParams& p) {
  const int K = p.c0 + p.c1, N = p.n_total;
  const int wtaps = p.nphase * p.ntaps;          // distinct kernel taps (phases partition them)
  for (int ph = 0; ph < p.nphase; ++ph)
    for (int t = 0; t < p.ntaps; ++t)
      if ((int)(int8_t)(p.tap[ph][t] & 0xff) == -128) return false;     // padded phases (odd kernels): direct kernel
  return N <= 2 && K % 16 == 0 && p.c0 % 16 == 0 && (wtaps * N) % 4 == 0 && wtaps * N <= 32 && !p.y_mask;
}

int64_t two_stage_bytes(const GatherGemmParams& p) {
  const int cols = p.nphase * p.ntaps * p.n_total;
  return (int64_t)sizeof(float) * p.batch * p.in_h * p.in_w * cols;
}

int run_two_stage(const GatherGemmParams& p, float* ws, hipStream_t stream, const char** name_only) {
  const int cols = p.nphase * p.ntaps * p.n_total;
  GatherGemmParams g = {};
  g.a0 = p.a0; g.a1 = p.a1; g.c0 = p.c0; g.c1 = p.c1; g.a0_pitch = p.a0_pitch; g.a1_pitch = p.a1_pitch;
  g.a_h = p.a_h; g.in_h = p.in_h; g.in_w = p.in_w;
  g.in_scale = p.in_scale; g.in_shift = p.in_shift; g.in_act = p.in_act;
  g.a_mask = p.a_mask; g.a_mask_scale = p.a_mask_scale;
  g.batch = p.batch; g.gh = p.in_h; g.gw = p.in_w; g.sy = g.sx = 1;
  g.nphase = 1; g.ntaps = 1; g.tap[0][0] = 0;
  g.w = p.w; g.n_total = 32; g.n_valid = cols; g.n_split = 32;
  g.osy = g.osx = 1;
  g.out_h = p.in_h; g.out_w = p.in_w;
  g.d[0].p = ws; g.d[0].pitch = p.in_w; g.d[0].c = cols;
  if (name_only) return launch_gather_gemm(g, /*b_kn=*/false, stream, name_only);
  int rc = launch_gather_gemm(g, /*b_kn=*/false, stream, nullptr);
  if (rc != ADVOC_OK) return rc;
  return launch_tap_sum(p, ws, cols, stream);
}

// every kernel indexes activations with 32-bit element offsets: refuse tensors beyond that range
bool fits_int32(int64_t batch, int64_t rows, int64_t pitch, int64_t c) {
  return batch * rows * pitch * c <= 0x7fffffffLL;
}

int run_gather(const GatherGemmParams& p, bool b_kn, hipStream_t stream, const char** name_only = nullptr,
               float* ws = nullptr, int64_t ws_bytes = 0) {
  const int K = p.c0 + p.c1, N = p.n_total;
  if (!fits_int32(p.batch, p.a_h, p.a0_pitch, p.c0) || !fits_int32(p.batch, p.a_h, p.a1_pitch, p.c1) ||
      !fits_int32(p.batch, p.out_h, p.d[0].pitch, p.d[0].c) || !fits_int32(p.batch, p.out_h, p.d[1].pitch, p.d[1].c))
    return ADVOC_ERR_UNSUPPORTED;
  if (two_stage_ok(p) && (b_kn ? N == 1 : true) && tuning().fused_taps && fused_taps_ok(p))
    return launch_fused_taps(p, stream, name_only);      // the two-stage path in one launch: S stays in LDS
  if (two_stage_ok(p) && ws && ws_bytes >= two_stage_bytes(p) && (b_kn ? N == 1 : true))
    return run_two_stage(p, ws, stream, name_only);
  if (K % 16 == 0 && p.c0 % 16 == 0 && N % 32 == 0 && p.n_split % 32 == 0) {
    // big launches: operand images + LDS-DMA kernel when the caller's workspace holds the images
    const int rc = launch_gather_gemm_h3(p, b_kn, stream, name_only, ws, ws_bytes, nullptr);
    if (rc != ADVOC_ERR_UNSUPPORTED) return rc;
    return launch_gather_gemm(p, b_kn, stream, name_only, ws, ws_bytes);
  }
  if (N <= 2 && K % 4 == 0 && p.c0 % 4 == 0) return launch_gather_dot(p, b_kn, stream, name_only);
  if (K <= 2 && N % 32 == 0 && p.ntaps * K <= 32) {
    const int rc = launch_thin_k_gemm(p, b_kn, stream, name_only);
    if (rc != ADVOC_ERR_UNSUPPORTED) return rc;     // tap span beyond the LDS patch: direct kernel below
  }
  if (K <= 2) return launch_gather_outer(p, b_kn, stream, name_only);
  return ADVOC_ERR_UNSUPPORTED;
}

void operand_from_inputs(const advoc_conv_layer* L, Operand& o) {
  o = Operand{};
  o.p0 = L->x0.p; o.c0 = L->x0.c; o.pitch0 = L->x0.w_pitch;
  o.p1 = L->x1.p; o.c1 = L->x1.p ? L->x1.c : 0; o.pitch1 = L->x1.p ? L->x1.w_pitch : 0;
  o.h = L->x0.h; o.w = L->x0.w;
  o.act = L->in_act; o.scale = L->in_scale; o.shift = L->in_shift;
  o.mask = L->in_mask; o.mask_scale = L->in_mask_scale;
}

void operand_from_dy(const advoc_conv_layer* L, const float* dy, Operand& o) {
  o = Operand{};
  o.p0 = dy; o.c0 = L->y.c; o.pitch0 = L->y.w_pitch;
  o.h = L->y.h; o.w = L->y.w;
  o.act = ADVOC_ACT_NONE;
  o.mask = L->drop_mask; o.mask_scale = L->drop_scale;
}

// dw has the layout of L->w.  conv: [tap][ci][co] -> rows a = ci (gathered input), cols b = co (dY
// at the output grid).  deconv: [tap][co][ci] -> rows a = co (dOut gathered at 2*iy - pad + k),
// cols b = ci (input at the input grid).  When the row operand would be wide but the column
// operand thin (conv with cout = 1, stride 1) the roles are swapped by walking the INPUT grid:
// dw[tap][ci][0] = sum_in x[in][ci] * dY[in + pad - k].
int build_backward_weight(const advoc_conv_layer* L, const float* dy, float* dw, WgradParams& p) {
  p = WgradParams{};
  if (L->kh * L->kw > kMaxTaps) return ADVOC_ERR_UNSUPPORTED;
  p.batch = L->x0.n;
  p.dw = dw;
  p.ntaps = L->kh * L->kw;
  const int cout = L->y.c;
  if (L->kind == ADVOC_DECONV) {
    operand_from_dy(L, dy, p.P);
    operand_from_inputs(L, p.Q);
    p.gh = L->x0.h; p.gw = L->x0.w;
    p.sy = L->sh; p.sx = L->sw;
    for (int ky = 0; ky < L->kh; ++ky)
      for (int kx = 0; kx < L->kw; ++kx)
        p.tap[ky * L->kw + kx] = pack_tap(ky - L->pad_t, kx - L->pad_l, ky * L->kw + kx);
    return ADVOC_OK;
  }
  if (cout <= 2 && cin_of(L) > 2) {
    if (L->sh != 1 || L->sw != 1 || cout != 1) return ADVOC_ERR_UNSUPPORTED;
    operand_from_dy(L, dy, p.P);
    operand_from_inputs(L, p.Q);
    p.gh = L->x0.h; p.gw = L->x0.w;
    p.sy = p.sx = 1;
    for (int ky = 0; ky < L->kh; ++ky)
      for (int kx = 0; kx < L->kw; ++kx)
        p.tap[ky * L->kw + kx] = pack_tap(L->pad_t - ky, L->pad_l - kx, ky * L->kw + kx);
    return ADVOC_OK;
  }
  operand_from_inputs(L, p.P);
  operand_from_dy(L, dy, p.Q);
  p.gh = L->y.h; p.gw = L->y.w;
  p.sy = L->sh; p.sx = L->sw;
  for (int ky = 0; ky < L->kh; ++ky)
    for (int kx = 0; kx < L->kw; ++kx)
      p.tap[ky * L->kw + kx] = pack_tap(ky - L->pad_t, kx - L->pad_l, ky * L->kw + kx);
  return ADVOC_OK;
}

}  // namespace
}  // namespace advoc

using namespace advoc;

extern "C" int advoc_conv_forward(const advoc_conv_layer* L, advoc_stream_t stream) {
  int rc = validate_layer(L);
  if (rc != ADVOC_OK) return rc;
  GatherGemmParams p;
  bool b_kn;
  rc = build_forward(L, p, b_kn);
  if (rc != ADVOC_OK) return rc;
  p.a_img_out = L->x_img; p.a_hdr_out = L->x_img ? L->x_hdr : nullptr;
  p.a_img_current = (L->img_flags & ADVOC_IMG_X_CURRENT) != 0;
  p.a_img_delayed = (L->img_flags & ADVOC_IMG_X_DELAYED) != 0;
  p.a_img_emitted = (L->img_flags & ADVOC_IMG_X_EMITTED) != 0;
  for (int k = 0; k < 2; ++k) {
    if (!L->y_img[k].img) continue;
    if (!L->y_img[k].hdr || L->y_img[k].act < ADVOC_ACT_NONE || L->y_img[k].act > ADVOC_ACT_RELU) return ADVOC_ERR_NULL;
    p.oimg[k].img = L->y_img[k].img;
    p.oimg[k].hdr = L->y_img[k].hdr;
    p.oimg[k].slope = L->y_img[k].act == ADVOC_ACT_LRELU02 ? 0.2f : (L->y_img[k].act == ADVOC_ACT_RELU ? 0.f : 1.f);
  }
  p.w_amax = L->w_amax;
  p.w_img = L->w_img[0]; p.w_img_hdr = L->w_img_hdr[0]; p.w_img_l1 = (L->img_flags & ADVOC_IMG_W_L1) != 0;
  p.a_img_bounded = (L->img_flags & ADVOC_IMG_X_BOUNDED) != 0;
  const int ymode = L->y_img[0].img ? L->y_img[0].mode : 0;
  if ((ymode & ADVOC_Y_IMAGE_ONLY) && !(ymode & ADVOC_Y_BOUNDED)) return ADVOC_ERR_UNSUPPORTED;
  if (ymode & ADVOC_Y_BOUNDED) {
    if (L->y_img[1].img) return ADVOC_ERR_UNSUPPORTED;
    p.oimg_bounded = 1;
    p.d0_no_store = (ymode & ADVOC_Y_IMAGE_ONLY) ? 1 : 0;
  }
  int emits = 0;
  if (p.oimg[0].img || p.oimg[1].img) {
    // (a launch outside the image kernels would silently ignore y_img and leave the consumers with stale images)
    GatherGemmParams q = p;
    const char* nm = nullptr;
    q.emit_report = &emits;
    rc = run_gather(q, b_kn, nullptr, &nm, L->workspace, L->workspace_bytes);
    if (rc != ADVOC_OK) return rc;
    if (!emits || (p.oimg_bounded && emits != 5)) return ADVOC_ERR_UNSUPPORTED;
  }
  const char* fwd_name = nullptr;
  if (p.oimg_bounded) {
    GatherGemmParams q = p;
    rc = run_gather(q, b_kn, nullptr, &fwd_name, L->workspace, L->workspace_bytes);
    if (rc != ADVOC_OK) return rc;
  }
  if (p.oimg_bounded && fwd_name && !strstr(fwd_name, "_h3_kernel")) {
    // max |input| for the bound: the inputs of the thin matrix kernel are fp32 tensors without an image (1-2 channels): a
    // magnitude pass over each source, into the reserved word 7 of the header the launch writes (the image kernels read the
    // magnitude of their input image from its header)
    if (L->x0.w_pitch != L->x0.w || (L->x1.p && L->x1.w_pitch != L->x1.w)) return ADVOC_ERR_UNSUPPORTED;
    // (r6) the bound is taken from the RAW inputs: an input affine (batch norm) or an input dropout mask (scale
    // > 1) the kernel applies on load would make it too small, and an image-only output has no fp32 tensor to fall back on
    if (L->in_scale || L->in_shift || L->in_mask) return ADVOC_ERR_UNSUPPORTED;
    unsigned* word = L->y_img[0].hdr + 7;
    hipError_t e = hipMemsetAsync(word, 0, 4, as_stream(stream));
    if (e != hipSuccess) { note_hip_error(e); return ADVOC_ERR_HIP; }
    rc = launch_amax_any(L->x0.p, (int64_t)L->x0.n * L->x0.h * L->x0.w_pitch * L->x0.c, word, as_stream(stream));
    if (rc == ADVOC_OK && L->x1.p)
      rc = launch_amax_any(L->x1.p, (int64_t)L->x1.n * L->x1.h * L->x1.w_pitch * L->x1.c, word, as_stream(stream));
    if (rc != ADVOC_OK) return rc;
    p.a_amax = word;
  }
  return run_gather(p, b_kn, as_stream(stream), nullptr, L->workspace, L->workspace_bytes);
}

extern "C" int advoc_conv_emits_images(const advoc_conv_layer* L) {
  if (validate_layer(L) != ADVOC_OK) return 0;
  GatherGemmParams p;
  bool b_kn;
  if (build_forward(L, p, b_kn) != ADVOC_OK) return 0;
  p.w_amax = L->w_amax;
  p.w_img = L->w_img[0]; p.w_img_hdr = L->w_img_hdr[0]; p.w_img_l1 = (L->img_flags & ADVOC_IMG_W_L1) != 0;
  p.a_img_out = L->x_img; p.a_hdr_out = L->x_img ? L->x_hdr : nullptr;
  int emits = 0;
  const char* nm = nullptr;
  p.emit_report = &emits;
  if (run_gather(p, b_kn, nullptr, &nm, L->workspace, L->workspace_bytes) != ADVOC_OK) return 0;
  if (!emits) return 0;
  // 2: also under the a-priori scale, fp32 tensor optional (ADVOC_Y_BOUNDED / ADVOC_Y_IMAGE_ONLY)
  GatherGemmParams q = p;
  int e2 = 0;
  q.emit_report = &e2;
  q.oimg_bounded = 1;
  q.oimg[0].img = reinterpret_cast<uint16_t*>(L->y.p); q.oimg[0].hdr = reinterpret_cast<unsigned*>(L->y.p);
  if (run_gather(q, b_kn, nullptr, &nm, L->workspace, L->workspace_bytes) == ADVOC_OK && e2 == 5) return 2;
  return 1;
}

extern "C" int advoc_conv_bias_fusable(const advoc_conv_layer* L) {
  if (validate_layer(L) != ADVOC_OK) return 0;
  return L->dy_img && L->dy_hdr && image_colsum_ok(L->y.c) ? 1 : 0;
}

extern "C" int advoc_conv_make_image(const advoc_conv_layer* L, int32_t which, const float* dy, advoc_stream_t stream) {
  int rc = validate_layer(L);
  if (rc != ADVOC_OK) return rc;
  if (which < 0 || which > 1 || (which == 1 && !dy)) return ADVOC_ERR_NULL;
  uint16_t* img = which == 0 ? L->x_img : L->dy_img;
  unsigned* hdr = which == 0 ? L->x_hdr : L->dy_hdr;
  if (!img || !hdr) return ADVOC_ERR_NULL;
  Operand o;
  if (which == 0) operand_from_inputs(L, o); else operand_from_dy(L, dy, o);
  if ((o.c0 % 32) || (o.c1 % 32)) return ADVOC_ERR_UNSUPPORTED;
  const bool delayed = (L->img_flags & (which == 0 ? ADVOC_IMG_X_DELAYED : ADVOC_IMG_DY_DELAYED)) != 0;
  // (the replica table of the bias-gradient sums sits where launch_gather_gemm_h3 keeps it: behind the 256-byte
  // header block of the workspace)
  float* table = L->workspace && L->workspace_bytes >= 256 + kColsumBytes
                     ? reinterpret_cast<float*>(reinterpret_cast<char*>(L->workspace) + 256) : nullptr;
  return wgrad_h3_make_image(o, L->x0.n, img, hdr, delayed, as_stream(stream), which == 1 ? L->db_fused : nullptr, table);
}

extern "C" int64_t advoc_conv_image_bytes(const advoc_conv_layer* L, int32_t which) {
  if (validate_layer(L) != ADVOC_OK || which < 0 || which > 1) return 0;
  WgradParams w;
  float dummy = 0.f;
  if (build_backward_weight(L, &dummy, &dummy, w) != ADVOC_OK || !wgrad_h3_eligible(w)) return 0;
  const bool p_is_inputs = w.P.p0 == L->x0.p;
  int64_t a, b;
  return wgrad_h3_operand_bytes((which == 0) == p_is_inputs ? w.P : w.Q, w.batch, &a, &b);
}

extern "C" int64_t advoc_conv_workspace_bytes(const advoc_conv_layer* L, int32_t direction) {
  if (validate_layer(L) != ADVOC_OK) return 0;
  if (direction == 2) {      // backward-weight: operand images of the image-based kernel
    WgradParams w;
    float dummy = 0.f;
    if (build_backward_weight(L, &dummy, &dummy, w) != ADVOC_OK) return 0;
    if (w.P.c0 + w.P.c1 <= 2) return 0;      // thin layers: no shared scratch (the bias table is advoc_conv_layer.wgrad_table)
    if (!wgrad_h3_eligible(w)) return 0;
    int64_t a, b, c, d;
    return 256 + wgrad_h3_operand_bytes(w.P, w.batch, &a, &b) + wgrad_h3_operand_bytes(w.Q, w.batch, &c, &d);
  }
  GatherGemmParams p;
  bool b_kn;
  float dummy = 0.f;
  const int rc = direction == 0 ? build_forward(L, p, b_kn)
                                : build_backward_data(L, &dummy, &dummy, L->x1.p ? &dummy : nullptr, 0, 0, p, b_kn);
  if (rc != ADVOC_OK) return 0;
  if (two_stage_ok(p) && !(b_kn && p.n_total != 1)) return two_stage_bytes(p);
  // gather-GEMM path: partial tiles of the tail split (igemm.hip, launch_cfg)
  const int K = p.c0 + p.c1, N = p.n_total;
  int64_t want = 0, want_img = 0;
  if (K % 16 == 0 && p.c0 % 16 == 0 && N % 32 == 0 && p.n_split % 32 == 0) {
    if (launch_gather_gemm_h3(p, b_kn, nullptr, nullptr, nullptr, 0, &want_img) != ADVOC_OK) want_img = 0;
    if (launch_gather_gemm(p, b_kn, nullptr, nullptr, nullptr, 0, &want) != ADVOC_OK) want = 0;
  }
  return want_img > want ? want_img : want;
}

namespace {
// does this backward-data call run on the kernel that can write advoc_conv_layer.dx_img (thin.hip: launch_thin_k_gemm with
// one destination, gating only)?
// 0: no; 2: the thin matrix kernel's backward-data instance (one-pass scale, dx0 only); 3: a patch kernel under the a-priori
// scale (ADVOC_DX_BOUNDED; dx1, when the layer has a second source, is an ordinary fp32 destination)
int dx_image_launch(const advoc_conv_layer* L, const float* dy, float* dx0, float* dx1, int accum0, int accum1) {
  if (!dx0 || accum1 || !tuning().emit_dx) return 0;
  if (accum0 && !L->dx_img.bound_add) return 0;        // an accumulating destination needs a bound of what it holds
  GatherGemmParams p;
  bool b_kn;
  if (build_backward_data(L, dy, dx0, dx1, accum0, 0, p, b_kn) != ADVOC_OK) return 0;
  p.obound_add = accum0 ? L->dx_img.bound_add : nullptr;
  if (p.y_mask || p.d[0].gmask || p.d[0].c % 64 || p.d[0].c > 1024 || !image_colsum_ok(p.d[0].c)) return 0;
  p.a_img_out = L->dy_img; p.a_hdr_out = L->dy_img ? L->dy_hdr : nullptr;
  p.w_amax = L->w_amax;
  p.w_img = L->w_img[1]; p.w_img_hdr = L->w_img_hdr[1]; p.w_img_l1 = (L->img_flags & ADVOC_IMG_W_L1) != 0;
  p.oimg[0].img = reinterpret_cast<uint16_t*>(dx0);       // (any non-null value: asks the launcher what it would do)
  p.oimg[0].hdr = reinterpret_cast<unsigned*>(dx0);
  int emits = 0;
  const char* nm = nullptr;
  p.emit_report = &emits;
  if (!accum0 && !dx1 && !L->x1.p && run_gather(p, b_kn, nullptr, &nm, L->workspace, L->workspace_bytes) == ADVOC_OK && emits == 2)
    return 2;
  emits = 0;
  p.oimg_bounded = 1;
  if (run_gather(p, b_kn, nullptr, &nm, L->workspace, L->workspace_bytes) != ADVOC_OK) return 0;
  // 4: the thin matrix kernel under the bound (it needs max |w| on the device: advoc_conv_layer.w_amax), a second source's
  // gradient as an ordinary fp32 destination
  return emits == 3 ? 3 : (emits == 2 ? 4 : 0);
}
}  // namespace

extern "C" int advoc_conv_backward_data(const advoc_conv_layer* L, const float* dy, float* dx0,
                                        float* dx1, int32_t accum0, int32_t accum1,
                                        advoc_stream_t stream) {
  int rc = validate_layer(L);
  if (rc != ADVOC_OK) return rc;
  if (!dy) return ADVOC_ERR_NULL;
  if (!dx0 && !(dx1 && L->x1.p)) return ADVOC_OK;
  GatherGemmParams p;
  bool b_kn;
  rc = build_backward_data(L, dy, dx0, dx1, accum0, accum1, p, b_kn);
  if (rc != ADVOC_OK) return rc;
  p.a_img_out = L->dy_img; p.a_hdr_out = L->dy_img ? L->dy_hdr : nullptr;
  p.a_img_current = (L->img_flags & ADVOC_IMG_DY_CURRENT) != 0;
  p.a_img_delayed = (L->img_flags & ADVOC_IMG_DY_DELAYED) != 0;
  p.a_img_emitted = (L->img_flags & ADVOC_IMG_DY_EMITTED) != 0;
  p.a_img_bounded = (L->img_flags & ADVOC_IMG_DY_BOUNDED) != 0;
  p.d1_amax_out = (dx1 && L->x1.p) ? L->dx1_amax : nullptr;
  if (L->img_flags & ADVOC_IMG_X_GATES) {
    // the fp32 inputs were never written: gate on the sign of x_img (source 1 behind source 0 at its 256-byte-rounded size)
    if (!L->x_img || L->in_scale || L->in_mask) return ADVOC_ERR_UNSUPPORTED;
    const int64_t b0 = ((int64_t)4 * L->x0.n * L->x0.h * L->x0.w_pitch * L->x0.c + 255) / 256 * 256;
    p.d[0].ximg = L->x_img;
    p.d[1].ximg = reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(L->x_img) + b0);
    const char* nm = nullptr;
    rc = run_gather(p, b_kn, nullptr, &nm, L->workspace, L->workspace_bytes);
    if (rc != ADVOC_OK) return rc;
    if (!nm || !strstr(nm, "patch_gemm_h3_kernel")) return ADVOC_ERR_UNSUPPORTED;
  }
  p.a_colsum = L->dy_img ? L->db_fused : nullptr;       // the bias gradient rides in the dy image pass (igemm_h3.hip)
  p.w_amax = L->w_amax;
  p.w_img = L->w_img[1]; p.w_img_hdr = L->w_img_hdr[1]; p.w_img_l1 = (L->img_flags & ADVOC_IMG_W_L1) != 0;
  if (L->dx_img.img) {
    // the lower layer's output-gradient image from this call's epilogue (advoc_conv_layer.dx_img)
    if (!L->dx_img.hdr || (L->dx_img.colsum && !L->dx_img.table)) return ADVOC_ERR_NULL;
    int kind = dx_image_launch(L, dy, dx0, dx1, accum0, accum1);
    const bool bounded = (L->dx_img.mode & ADVOC_DX_BOUNDED) != 0;
    // (2: the thin kernel under the one-pass scale, fp32 dx0 stored; 3 / 4: a patch kernel / the thin kernel under the bound;
    // a launch that reports 2 can also run as 4 when max |w| is on the device)
    if (kind == 2 && bounded) {
      GatherGemmParams q = p;
      q.oimg_bounded = 1; q.oimg[0].img = L->dx_img.img; q.oimg[0].hdr = L->dx_img.hdr;
      int emits = 0;
      const char* nm = nullptr;
      q.emit_report = &emits;
      kind = run_gather(q, b_kn, nullptr, &nm, L->workspace, L->workspace_bytes) == ADVOC_OK && emits == 2 ? 4 : 0;
    }
    if (kind == 0 || (kind == 2 && (L->dx_img.mode & ADVOC_DX_IMAGE_ONLY)) || (kind >= 3 && !bounded)) return ADVOC_ERR_UNSUPPORTED;
    p.oimg[0].img = L->dx_img.img;
    p.oimg[0].hdr = L->dx_img.hdr;
    p.oimg[0].slope = 1.f;
    p.ocolsum_out = L->dx_img.colsum;
    p.ocolsum_table = L->dx_img.table;
    if (kind >= 3) {
      p.oimg_bounded = 1;
      p.obound_add = L->dx_img.bound_add;
      p.d0_no_store = (L->dx_img.mode & ADVOC_DX_IMAGE_ONLY) ? 1 : 0;
    }
    if (kind == 4) {
      // the thin operand is an fp32 tensor without an image: its largest magnitude by a pass of its own (1-2 channels), into
      // the reserved word 7 of the header the launch writes
      if (L->y.w_pitch != L->y.w) return ADVOC_ERR_UNSUPPORTED;     // (columns beyond the logical width hold anything)
      unsigned* word = L->dx_img.hdr + 7;
      hipError_t e = hipMemsetAsync(word, 0, 4, as_stream(stream));
      if (e != hipSuccess) { note_hip_error(e); return ADVOC_ERR_HIP; }
      const int64_t elems = (int64_t)L->y.n * L->y.h * L->y.w_pitch * L->y.c;
      rc = launch_amax_any(dy, elems, word, as_stream(stream));
      if (rc != ADVOC_OK) return rc;
      p.a_amax = word;
    }
  }
  return run_gather(p, b_kn, as_stream(stream), nullptr, L->workspace, L->workspace_bytes);
}

extern "C" int advoc_conv_emits_dx_image(const advoc_conv_layer* L) {
  if (validate_layer(L) != ADVOC_OK) return 0;
  float dummy = 0.f;
  // (with dx_img.bound_add set: as an ACCUMULATING call, accum0)
  return dx_image_launch(L, &dummy, &dummy, L->x1.p ? &dummy : nullptr, L->dx_img.bound_add ? 1 : 0, 0);
}

extern "C" int advoc_conv_gates_on_image(const advoc_conv_layer* L) {
  if (validate_layer(L) != ADVOC_OK || !L->x_img || L->in_scale || L->in_mask || L->in_act == ADVOC_ACT_NONE) return 0;
  GatherGemmParams p;
  bool b_kn;
  float dummy = 0.f;
  if (build_backward_data(L, &dummy, &dummy, L->x1.p ? &dummy : nullptr, 0, 0, p, b_kn) != ADVOC_OK) return 0;
  p.a_img_out = L->dy_img; p.a_hdr_out = L->dy_img ? L->dy_hdr : nullptr;
  p.w_amax = L->w_amax;
  p.w_img = L->w_img[1]; p.w_img_hdr = L->w_img_hdr[1]; p.w_img_l1 = (L->img_flags & ADVOC_IMG_W_L1) != 0;
  p.d[0].ximg = L->x_img; p.d[1].ximg = L->x_img;
  const char* nm = nullptr;
  if (run_gather(p, b_kn, nullptr, &nm, L->workspace, L->workspace_bytes) != ADVOC_OK) return 0;
  return nm && strstr(nm, "patch_gemm_h3_kernel") ? 1 : 0;
}

extern "C" int64_t advoc_conv_wgrad_ws_bytes(const advoc_conv_layer* L) {
  if (validate_layer(L) != ADVOC_OK) return 0;
  WgradParams p;
  float dummy = 0.f;
  if (build_backward_weight(L, &dummy, &dummy, p) != ADVOC_OK) return 0;
  return wgrad_h3_partial_bytes(p);
}

extern "C" int advoc_conv_backward_weight(const advoc_conv_layer* L, const float* dy, float* dw,
                                          float* db, int32_t accumulate, advoc_stream_t stream) {
  int rc = validate_layer(L);
  if (rc != ADVOC_OK) return rc;
  if (!dy || !dw) return ADVOC_ERR_NULL;
  WgradParams p;
  rc = build_backward_weight(L, dy, dw, p);
  if (rc != ADVOC_OK) return rc;
  if (!fits_int32(L->x0.n, L->x0.h, L->x0.w_pitch, L->x0.c) ||
      !fits_int32(L->x0.n, L->x0.h, L->x1.p ? L->x1.w_pitch : 0, L->x1.p ? L->x1.c : 0) ||
      !fits_int32(L->y.n, L->y.h, L->y.w_pitch, L->y.c))
    return ADVOC_ERR_UNSUPPORTED;
  p.accumulate = accumulate;
  p.part_ws = L->wgrad_ws;
  p.part_ws_bytes = L->wgrad_ws ? L->wgrad_ws_bytes : 0;
  const int ca = p.P.c0 + p.P.c1;
  // (r6) an image-only operand that is not flagged current cannot be rebuilt (there is no fp32 tensor to rebuild it from)
  if ((L->img_flags & ADVOC_IMG_X_GATES) && !(L->img_flags & ADVOC_IMG_X_CURRENT)) return ADVOC_ERR_UNSUPPORTED;
  if ((L->img_flags & ADVOC_IMG_DY_BOUNDED) && !(L->img_flags & ADVOC_IMG_DY_CURRENT)) return ADVOC_ERR_UNSUPPORTED;
  if (ca <= 2 && (L->img_flags & (ADVOC_IMG_X_GATES | ADVOC_IMG_DY_BOUNDED))) return ADVOC_ERR_UNSUPPORTED;
  if (ca <= 2) {
    const int cb = p.Q.c0 + p.Q.c1;
    // the bias gradient rides in the weight-gradient kernel when its wide operand IS the output gradient (encoder_1,
    // layer_1): that kernel reads every dy element exactly once anyway
    const bool thin_mfma = cb % 32 == 0 && p.ntaps * ca <= 32;
    bool db_fused = false;
    if (thin_mfma && db && p.Q.p0 == dy && !p.Q.p1 && cb <= 1024 && L->wgrad_table && tuning().thin_wgrad_bias) {
      if (!accumulate) {
        hipError_t e = hipMemsetAsync(db, 0, sizeof(float) * (size_t)cb, as_stream(stream));
        if (e != hipSuccess) { note_hip_error(e); return ADVOC_ERR_HIP; }
      }
      p.qsum_out = db;
      p.qsum_table = L->wgrad_table;      // the layer's own: this call may run beside workspace users on another stream
      db_fused = true;
    }
    rc = thin_mfma ? launch_wgrad_thin_mfma(p, as_stream(stream)) : ADVOC_ERR_UNSUPPORTED;
    if (rc == ADVOC_ERR_UNSUPPORTED) {                 // odd channel counts / wide tap spans
      p.qsum_out = p.qsum_table = nullptr;
      db_fused = false;
      rc = launch_wgrad_thin(p, as_stream(stream));
    }
    if (db_fused) db = nullptr;
  } else {
    rc = ADVOC_ERR_UNSUPPORTED;
    if (wgrad_h3_eligible(p)) {
      // operand images: the layer's persistent ones when they are current (left by this step's forward /
      // backward-data call), else made here -- into the persistent buffer if there is one, else into the workspace
      // ([P header 128 B][Q header 128 B][P image][Q image])
      int64_t pb0, pb1, qb0, qb1;
      const int64_t pbytes = wgrad_h3_operand_bytes(p.P, p.batch, &pb0, &pb1);
      const int64_t qbytes = wgrad_h3_operand_bytes(p.Q, p.batch, &qb0, &qb1);
      const bool p_is_inputs = p.P.p0 == L->x0.p;
      struct Home { uint16_t* img; unsigned* hdr; bool current; bool delayed; };
      const Home in_home = {L->x_img, L->x_img ? L->x_hdr : nullptr, (L->img_flags & ADVOC_IMG_X_CURRENT) != 0,
                            (L->img_flags & ADVOC_IMG_X_DELAYED) != 0};
      const Home dy_home = {L->dy_img, L->dy_img ? L->dy_hdr : nullptr, (L->img_flags & ADVOC_IMG_DY_CURRENT) != 0,
                            (L->img_flags & ADVOC_IMG_DY_DELAYED) != 0};
      Home hp = p_is_inputs ? in_home : dy_home, hq = p_is_inputs ? dy_home : in_home;
      int64_t ws_need = 256 + (hp.img ? 0 : pbytes) + (hq.img ? 0 : qbytes);
      if ((hp.img && hq.img) || (L->workspace && L->workspace_bytes >= ws_need)) {
        char* ws = reinterpret_cast<char*>(L->workspace);
        if (!hp.img) { hp = {reinterpret_cast<uint16_t*>(ws + 256), reinterpret_cast<unsigned*>(ws), false, false}; }
        if (!hq.img) {
          hq = {reinterpret_cast<uint16_t*>(ws + 256 + (hp.img == reinterpret_cast<uint16_t*>(ws + 256) ? pbytes : 0)),
                reinterpret_cast<unsigned*>(ws + 128), false, false};
        }
        rc = ADVOC_OK;
        if (!tuning().h3_skip_prep) {      // (micro-benchmarks reuse the images of the previous call)
          if (!hp.current) rc = wgrad_h3_make_image(p.P, p.batch, hp.img, hp.hdr, hp.delayed, as_stream(stream));
          if (rc == ADVOC_OK && !hq.current) rc = wgrad_h3_make_image(p.Q, p.batch, hq.img, hq.hdr, hq.delayed, as_stream(stream));
        }
        if (rc == ADVOC_OK) rc = launch_wgrad_h3(p, hp.img, hp.hdr, hq.img, hq.hdr, as_stream(stream));
      }
    }
    // (r6) operands that exist as images only -- ADVOC_IMG_X_GATES: the inputs, ADVOC_IMG_DY_BOUNDED: the output
    // gradient, both never written as fp32 -- can be read by the image kernel alone: no fall-back to the fp32 kernels
    if (rc == ADVOC_ERR_UNSUPPORTED && (L->img_flags & (ADVOC_IMG_X_GATES | ADVOC_IMG_DY_BOUNDED))) return rc;
    if (rc == ADVOC_ERR_UNSUPPORTED) rc = launch_wgrad_mfma(p, as_stream(stream));
  }
  if (rc != ADVOC_OK) return rc;
  // (the bias gradient below reads the fp32 dy: refused when that tensor was never written)
  if (db && (L->img_flags & ADVOC_IMG_DY_BOUNDED)) return ADVOC_ERR_UNSUPPORTED;
  if (db)
    rc = launch_bias_grad(dy, L->drop_mask, L->drop_scale, (int64_t)L->y.n * L->y.h, L->y.w,
                          L->y.w_pitch, L->y.c, db, accumulate, as_stream(stream));
  return rc;
}

extern "C" int advoc_conv_backward_bias(const advoc_conv_layer* L, const float* dy, float* db,
                                        int32_t accumulate, advoc_stream_t stream) {
  const int rc = validate_layer(L);
  if (rc != ADVOC_OK) return rc;
  if (!dy || !db) return ADVOC_ERR_NULL;
  return launch_bias_grad(dy, L->drop_mask, L->drop_scale, (int64_t)L->y.n * L->y.h, L->y.w,
                          L->y.w_pitch, L->y.c, db, accumulate, as_stream(stream));
}

// {taps, n_total, ktot, b_kn, bytes} of the weight image direction 0 (forward) / 1 (backward-data) of this layer reads;
// bytes = 0 when that direction does not run on the image kernels
extern "C" int advoc_conv_weight_image_desc(const advoc_conv_layer* L, int32_t direction, int64_t* out5_host) {
  int rc = validate_layer(L);
  if (rc != ADVOC_OK) return rc;
  if (!out5_host) return ADVOC_ERR_NULL;
  if (direction != 0 && direction != 1) return ADVOC_ERR_UNSUPPORTED;
  for (int i = 0; i < 5; ++i) out5_host[i] = 0;
  GatherGemmParams p;
  bool b_kn;
  float dummy = 0.f;
  rc = direction == 0 ? build_forward(L, p, b_kn)
                      : build_backward_data(L, &dummy, &dummy, L->x1.p ? &dummy : nullptr, 0, 0, p, b_kn);
  if (rc != ADVOC_OK) return rc;
  const char* name = nullptr;
  rc = run_gather(p, b_kn, nullptr, &name, L->workspace, L->workspace_bytes);
  if (rc != ADVOC_OK) return rc;
  int taps, n_total, ktot;
  if (!name || !strstr(name, "_h3_kernel") || !h3_weight_image_shape(p, &taps, &n_total, &ktot)) return ADVOC_OK;
  out5_host[0] = taps; out5_host[1] = n_total; out5_host[2] = ktot; out5_host[3] = b_kn ? 1 : 0;
  out5_host[4] = ((int64_t)4 * taps * n_total * ktot + 255) / 256 * 256;
  return ADVOC_OK;
}

extern "C" int advoc_conv_kernel_name(const advoc_conv_layer* L, int32_t direction, char* buf_host,
                                      int32_t buf_len) {
  int rc = validate_layer(L);
  if (rc != ADVOC_OK) return rc;
  if (!buf_host || buf_len < 2) return ADVOC_ERR_NULL;
  const char* name = nullptr;
  if (direction == 0 || direction == 1) {
    GatherGemmParams p;
    bool b_kn;
    float dummy = 0.f;
    rc = direction == 0 ? build_forward(L, p, b_kn)
                        : build_backward_data(L, &dummy, &dummy, L->x1.p ? &dummy : nullptr, 0, 0, p, b_kn);
    if (rc != ADVOC_OK) return rc;
    rc = run_gather(p, b_kn, nullptr, &name, L->workspace, L->workspace_bytes);
  } else if (direction == 2) {
    WgradParams p;
    float dummy = 0.f;
    rc = build_backward_weight(L, &dummy, &dummy, p);
    if (rc != ADVOC_OK) return rc;
    const int ca = p.P.c0 + p.P.c1, cb = p.Q.c0 + p.Q.c1;
    if (ca <= 2)
      rc = (cb % 32 == 0 && p.ntaps * ca <= 32) ? launch_wgrad_thin_mfma(p, nullptr, &name)
                                                : launch_wgrad_thin(p, nullptr, &name);
    else {
      int64_t a, b, c, d;
      rc = ADVOC_ERR_UNSUPPORTED;
      if (wgrad_h3_eligible(p) && L->workspace &&
          L->workspace_bytes >= 256 + wgrad_h3_operand_bytes(p.P, p.batch, &a, &b) + wgrad_h3_operand_bytes(p.Q, p.batch, &c, &d))
        rc = launch_wgrad_h3(p, nullptr, nullptr, nullptr, nullptr, nullptr, &name);
      if (rc == ADVOC_ERR_UNSUPPORTED) rc = launch_wgrad_mfma(p, nullptr, &name);
    }
  } else {
    return ADVOC_ERR_UNSUPPORTED;
  }
  if (rc != ADVOC_OK) return rc;
  int i = 0;
  for (; name[i] && i < buf_len - 1; ++i) buf_host[i] = name[i];
  buf_host[i] = 0;
  return ADVOC_OK;
}
