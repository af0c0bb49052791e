// Direct kernels for the thin edge layers of the AdVoc nets (gfx950): 1-2 input channels
// (generator encoder_1, discriminator layer_1) or a single output channel (generator decoder_1,
// discriminator layer_5).  Arithmetic intensity is 7-14 flop/byte: these
// are HBM-bound, so no MFMA -- coalesced channel-contiguous accesses, operands broadcast from
// L1, wave-shuffle reductions.  They consume the same GatherGemmParams as the MFMA kernel.
//
// Reference ops replaced: Conv2D / Conv2DBackpropInput for advoc_model.py:91-94 (encoder_1),
// :153-158 (decoder_1), :185-188 (layer_1), :199-202 (layer_5) and their gradients.
#include "conv_internal.h"

namespace advoc {
namespace {

__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == ADVOC_ACT_LRELU02) return fmaxf(0.2f * v, v);
  if (act == ADVOC_ACT_RELU) return fmaxf(v, 0.f);
  return v;
}

__device__ __forceinline__ float act_bwd(float x, int act) {
  if (act == ADVOC_ACT_LRELU02) return x > 0.f ? 1.f : 0.2f;   // TF MaximumGrad: tie -> alpha branch
  if (act == ADVOC_ACT_RELU) return x > 0.f ? 1.f : 0.f;
  return 1.f;
}

struct GridPoint {
  int img, gy, gx;
};

__device__ __forceinline__ GridPoint decompose(int64_t m, int gh, int gw) {
  GridPoint g;
  g.gx = (int)(m % gw);
  const int64_t t = m / gw;
  g.gy = (int)(t % gh);
  g.img = (int)(t / gh);
  return g;
}

// Epilogue shared by both kernels: value v of output channel n at grid point g.
__device__ __forceinline__ void store_result(const GatherGemmParams& p, int phase, const GridPoint& g,
                                             int n, float v) {
  const int oy = g.gy * p.osy + p.ooy[phase], ox = g.gx * p.osx + p.oox[phase];
  if (oy >= p.out_h || ox >= p.out_w) return;
  const int di = n >= p.n_split ? 1 : 0;
  const GemmDest& d = p.d[di];
  if (d.p == nullptr) return;
  const int ch = di ? n - p.n_split : n;
  const int64_t off = (((int64_t)g.img * p.out_h + oy) * d.pitch + ox) * d.c + ch;
  if (p.bias) v += p.bias[n];
  if (p.y_mask) v *= p.y_mask[off] * p.y_mask_scale;
  if (p.grad_act != ADVOC_ACT_NONE) {
    float x = d.xpre[off];
    if (d.gscale) x = x * d.gscale[ch] + d.gshift[ch];
    v *= act_bwd(x, p.grad_act);
  }
  if (d.gmask) v *= d.gmask[off] * d.gmask_scale;
  if (d.accum) v += d.p[off];
  d.p[off] = v;
}

// ---------------------------------------------------------------------------------------------
// gather_dot: few outputs (N <= 2), wide K.  G lanes cooperate on one grid point.
// ---------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void gather_dot_kernel(const GatherGemmParams p) {
  __shared__ int s_tap[kMaxPhases][kMaxTaps];
  if (threadIdx.x < kMaxPhases * kMaxTaps) s_tap[threadIdx.x / kMaxTaps][threadIdx.x % kMaxTaps] =
      p.tap[threadIdx.x / kMaxTaps][threadIdx.x % kMaxTaps];
  __syncthreads();

  constexpr int PTS = 256 / G;  // grid points per block
  const int sub = threadIdx.x % G;
  const int64_t M = (int64_t)p.batch * p.gh * p.gw;
  const int64_t m = (int64_t)blockIdx.x * PTS + threadIdx.x / G;
  const bool live = m < M;
  const GridPoint g = decompose(live ? m : 0, p.gh, p.gw);
  const int ktot = p.c0 + p.c1;
  const int N = p.n_total;
  const float slope = p.in_act == ADVOC_ACT_LRELU02 ? 0.2f : (p.in_act == ADVOC_ACT_RELU ? 0.f : 1.f);

  // All sub-pixel phases of a grid point are computed by the same lanes back to back: the 2x2-tap
  // windows of the four phases overlap (9 distinct input pixels for 16 tap reads), so the
  // re-reads hit this CU's L1 instead of going back to L2 from four different workgroups.
  for (int phase = 0; phase < p.nphase; ++phase) {
    float acc0 = 0.f, acc1 = 0.f;
    if (live) {
      for (int t = 0; t < p.ntaps; ++t) {
        const int tp = s_tap[phase][t];
        const int iy = g.gy * p.sy + (int)(int8_t)(tp & 0xff);
        const int ix = g.gx * p.sx + (int)(int8_t)((tp >> 8) & 0xff);
        if ((unsigned)iy >= (unsigned)p.in_h || (unsigned)ix >= (unsigned)p.in_w) continue;
        const int wtap = tp >> 16;
        for (int k = 4 * sub; k < ktot; k += 4 * G) {
          const bool second = k >= p.c0;
          const float* src = second ? p.a1 : p.a0;
          const int cs = second ? p.c1 : p.c0;
          const int pitch = second ? p.a1_pitch : p.a0_pitch;
          const int64_t off = (((int64_t)g.img * p.a_h + iy) * pitch + ix) * cs + (second ? k - p.c0 : k);
          float4 v = *reinterpret_cast<const float4*>(src + off);
          if (p.in_scale) {
            const float4 sc = *reinterpret_cast<const float4*>(p.in_scale + k);
            const float4 sh = *reinterpret_cast<const float4*>(p.in_shift + k);
            v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
          }
          v.x = fmaxf(v.x, slope * v.x); v.y = fmaxf(v.y, slope * v.y);
          v.z = fmaxf(v.z, slope * v.z); v.w = fmaxf(v.w, slope * v.w);
          if (p.a_mask && !second) {
            const uchar4 mk = *reinterpret_cast<const uchar4*>(p.a_mask + off);
            v.x *= mk.x * p.a_mask_scale; v.y *= mk.y * p.a_mask_scale;
            v.z *= mk.z * p.a_mask_scale; v.w *= mk.w * p.a_mask_scale;
          }
          // weights: N == 1 -> w[wtap*K + k]; N == 2 only in the [tap][N][K] layout
          const float4 w0 = *reinterpret_cast<const float4*>(p.w + ((int64_t)wtap * N) * ktot + k);
          acc0 = fmaf(v.x, w0.x, fmaf(v.y, w0.y, fmaf(v.z, w0.z, fmaf(v.w, w0.w, acc0))));
          if (N > 1) {
            const float4 w1 = *reinterpret_cast<const float4*>(p.w + ((int64_t)wtap * N + 1) * ktot + k);
            acc1 = fmaf(v.x, w1.x, fmaf(v.y, w1.y, fmaf(v.z, w1.z, fmaf(v.w, w1.w, acc1))));
          }
        }
      }
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
      acc0 += __shfl_xor(acc0, o, 64);
      acc1 += __shfl_xor(acc1, o, 64);
    }
    if (live && sub == 0) {
      store_result(p, phase, g, 0, acc0);
      if (N > 1) store_result(p, phase, g, 1, acc1);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// gather_outer: K <= 2 input channels, wide N.  One thread per (grid point, output channel).
// ---------------------------------------------------------------------------------------------
template <bool B_KN>
__global__ __launch_bounds__(256) void gather_outer_kernel(const GatherGemmParams p) {
  __shared__ int s_tap[kMaxTaps];
  const int phase = blockIdx.z;
  if (threadIdx.x < kMaxTaps) s_tap[threadIdx.x] = p.tap[phase][threadIdx.x];
  __syncthreads();

  const int N = p.n_total;
  const int64_t M = (int64_t)p.batch * p.gh * p.gw;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t m = idx / N;
  const int n = (int)(idx - m * N);
  if (m >= M) return;
  const GridPoint g = decompose(m, p.gh, p.gw);
  const int ktot = p.c0 + p.c1;

  float acc = 0.f;
  for (int t = 0; t < p.ntaps; ++t) {
    const int tp = s_tap[t];
    const int iy = g.gy * p.sy + (int)(int8_t)(tp & 0xff);
    const int ix = g.gx * p.sx + (int)(int8_t)((tp >> 8) & 0xff);
    if ((unsigned)iy >= (unsigned)p.in_h || (unsigned)ix >= (unsigned)p.in_w) continue;
    const int wtap = tp >> 16;
    for (int k = 0; k < ktot; ++k) {
      const bool second = k >= p.c0;
      const float* src = second ? p.a1 : p.a0;
      const int cs = second ? p.c1 : p.c0;
      const int pitch = second ? p.a1_pitch : p.a0_pitch;
      const int64_t off = (((int64_t)g.img * p.a_h + iy) * pitch + ix) * cs + (second ? k - p.c0 : k);
      float v = src[off];
      if (p.in_scale) v = v * p.in_scale[k] + p.in_shift[k];
      v = act_fwd(v, p.in_act);
      if (p.a_mask && !second) v *= p.a_mask[off] * p.a_mask_scale;
      const float w = B_KN ? p.w[((int64_t)wtap * ktot + k) * N + n] : p.w[((int64_t)wtap * N + n) * ktot + k];
      acc = fmaf(v, w, acc);
    }
  }
  store_result(p, phase, g, n, acc);
}

// ---------------------------------------------------------------------------------------------
// tap_sum: out[g, n] = bias[n] + sum_t S[g*s + d_t][wtap_t * N + n]
// A workgroup owns a TY x TX patch of grid points of one image.  The S rows/columns its taps can
// reach are staged in LDS ONCE with coalesced 16-byte loads (zeros outside the image), pixel
// stride SC + 1 words so that the per-lane gathers are bank-conflict free; every S value is used
// by exactly one (phase, tap) of one grid point, so HBM sees S once.
// ---------------------------------------------------------------------------------------------
constexpr int kTsTX = 64, kTsTY = 4;

__global__ __launch_bounds__(256) void tap_sum_kernel(const GatherGemmParams p, const float* __restrict__ S,
                                                      int sc, int dy_min, int dx_min, int ry, int rx,
                                                      int tiles_y, int tiles_x) {
  extern __shared__ __attribute__((aligned(16))) float s_patch[];   // [ry][rx][sc + 1]
  __shared__ int s_tap[kMaxPhases][kMaxTaps];
  if (threadIdx.x < kMaxPhases * kMaxTaps) s_tap[threadIdx.x / kMaxTaps][threadIdx.x % kMaxTaps] =
      p.tap[threadIdx.x / kMaxTaps][threadIdx.x % kMaxTaps];
  int b = blockIdx.x;
  const int tx_i = b % tiles_x; b /= tiles_x;
  const int ty_i = b % tiles_y;
  const int img = b / tiles_y;
  const int gy0 = ty_i * kTsTY, gx0 = tx_i * kTsTX;
  const int iy0 = gy0 * p.sy + dy_min, ix0 = gx0 * p.sx + dx_min;
  const int ld = sc + 1;
  const int quads = sc / 4;
  // stage: one float4 of one pixel per thread per step
  for (int i = threadIdx.x; i < ry * rx * quads; i += 256) {
    const int q = i % quads;
    const int pix = i / quads;
    const int px = pix % rx, py = pix / rx;
    const int iy = iy0 + py, ix = ix0 + px;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)iy < (unsigned)p.in_h && (unsigned)ix < (unsigned)p.in_w)
      v = *reinterpret_cast<const float4*>(S + (((int64_t)img * p.in_h + iy) * p.in_w + ix) * sc + 4 * q);
    float* dst = s_patch + pix * ld + 4 * q;
    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
  }
  __syncthreads();
  GridPoint g;
  g.img = img;
  g.gy = gy0 + threadIdx.x / kTsTX;
  g.gx = gx0 + threadIdx.x % kTsTX;
  if (g.gy >= p.gh || g.gx >= p.gw) return;
  const int ly = (g.gy - gy0) * p.sy - dy_min, lx = (g.gx - gx0) * p.sx - dx_min;
  const int N = p.n_total;
  for (int phase = 0; phase < p.nphase; ++phase) {
    float acc0 = 0.f, acc1 = 0.f;
    for (int t = 0; t < p.ntaps; ++t) {
      const int tp = s_tap[phase][t];
      const int py = ly + (int)(int8_t)(tp & 0xff);
      const int px = lx + (int)(int8_t)((tp >> 8) & 0xff);
      const float* row = s_patch + (py * rx + px) * ld + (tp >> 16) * N;
      acc0 += row[0];
      if (N > 1) acc1 += row[1];
    }
    store_result(p, phase, g, 0, acc0);
    if (N > 1) store_result(p, phase, g, 1, acc1);
  }
}

}  // namespace

int launch_tap_sum(const GatherGemmParams& p, const float* S, int s_channels, hipStream_t stream,
                   const char** name_only) {
  if (name_only) { *name_only = "tap_sum_kernel"; return ADVOC_OK; }
  if (s_channels % 4) return ADVOC_ERR_UNSUPPORTED;
  int dy_min = 127, dy_max = -128, dx_min = 127, dx_max = -128;
  for (int ph = 0; ph < p.nphase; ++ph)
    for (int t = 0; t < p.ntaps; ++t) {
      const int dy = (int)(int8_t)(p.tap[ph][t] & 0xff), dx = (int)(int8_t)((p.tap[ph][t] >> 8) & 0xff);
      dy_min = dy < dy_min ? dy : dy_min; dy_max = dy > dy_max ? dy : dy_max;
      dx_min = dx < dx_min ? dx : dx_min; dx_max = dx > dx_max ? dx : dx_max;
    }
  const int ry = (kTsTY - 1) * p.sy + (dy_max - dy_min) + 1;
  const int rx = (kTsTX - 1) * p.sx + (dx_max - dx_min) + 1;
  const size_t lds = sizeof(float) * (size_t)ry * rx * (s_channels + 1);
  if (lds > 150 * 1024) return ADVOC_ERR_UNSUPPORTED;
  const int tiles_y = (int)ceil_div(p.gh, kTsTY), tiles_x = (int)ceil_div(p.gw, kTsTX);
  const int64_t blocks = (int64_t)p.batch * tiles_y * tiles_x;
  if (blocks > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
  if (lds > 64 * 1024) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(tap_sum_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (attr != hipSuccess) { note_hip_error(attr); return ADVOC_ERR_HIP; }
  }
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(tap_sum_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, p, S, s_channels, dy_min,
                     dx_min, ry, rx, tiles_y, tiles_x);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

int launch_gather_dot(const GatherGemmParams& p, bool b_kn, hipStream_t stream, const char** name_only) {
  const int ktot = p.c0 + p.c1;
  if (p.n_total < 1 || p.n_total > 2 || ktot % 4 || p.c0 % 4) return ADVOC_ERR_UNSUPPORTED;
  if (p.n_total == 2 && b_kn) return ADVOC_ERR_UNSUPPORTED;
  const int64_t M = (int64_t)p.batch * p.gh * p.gw;
  int G = 8;
  while (G < 64 && G * 4 < ktot) G *= 2;
  if (name_only) {
    *name_only = G == 8 ? "gather_dot_kernel<8>" : G == 16 ? "gather_dot_kernel<16>"
                 : G == 32 ? "gather_dot_kernel<32>" : "gather_dot_kernel<64>";
    return ADVOC_OK;
  }
  const int64_t blocks = ceil_div(M, 256 / G);
  if (blocks > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
  dim3 grid((unsigned)blocks, 1, 1);      // phases are looped inside the kernel
  ADVOC_CLEAR_LAUNCH_ERROR();
  switch (G) {
    case 8: hipLaunchKernelGGL(gather_dot_kernel<8>, grid, dim3(256), 0, stream, p); break;
    case 16: hipLaunchKernelGGL(gather_dot_kernel<16>, grid, dim3(256), 0, stream, p); break;
    case 32: hipLaunchKernelGGL(gather_dot_kernel<32>, grid, dim3(256), 0, stream, p); break;
    default: hipLaunchKernelGGL(gather_dot_kernel<64>, grid, dim3(256), 0, stream, p); break;
  }
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

int launch_gather_outer(const GatherGemmParams& p, bool b_kn, hipStream_t stream, const char** name_only) {
  const int ktot = p.c0 + p.c1;
  if (ktot < 1 || ktot > 2) return ADVOC_ERR_UNSUPPORTED;
  if (name_only) {
    *name_only = b_kn ? "gather_outer_kernel<true>" : "gather_outer_kernel<false>";
    return ADVOC_OK;
  }
  const int64_t M = (int64_t)p.batch * p.gh * p.gw;
  const int64_t blocks = ceil_div(M * p.n_total, 256);
  if (blocks > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
  dim3 grid((unsigned)blocks, 1, (unsigned)p.nphase);
  ADVOC_CLEAR_LAUNCH_ERROR();
  if (b_kn) hipLaunchKernelGGL(gather_outer_kernel<true>, grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(gather_outer_kernel<false>, grid, dim3(256), 0, stream, p);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

}  // namespace advoc
