// Thin-layer kernels on the fp32 matrix cores, operands straight from global memory (gfx950).
//
// The four edge layers of the AdVoc nets have a contraction axis of only taps x channels <= 32
// (encoder_1: 16 x 1, discriminator layer_1: 16 x 2, decoder_1 / layer_5 backward-data: 16 x 1)
// next to a wide pixel axis.  They are HBM-bound (7-14 flop/byte): each activation
// byte is touched once.  The 32x32x2 fp32 MFMA does the few MACs per byte in ~1/64 of the
// instruction slots a scalar FMA loop needs, its C layout stores 128 B channel rows, and no LDS
// staging is needed because no operand is reused across waves.
//
//   thin_k_gemm_kernel : y[m, n] = sum_{k' < taps*K} A[m, k'] W[k', n]     (forward of 1-2 channel
//                        inputs; backward-data of 1-channel outputs).  One wave = 32 grid points x
//                        NT 32-channel tiles; weights live in registers for the life of the wave.
//   thin_wgrad_kernel  : dw[(tap, a), b] = sum_g P[g @ tap][a] Q[g][b]    (weight gradient when the
//                        gathered operand has 1-2 channels).  One wave streams a range of grid
//                        points, 2 per MFMA; Q rows are 128 B coalesced loads; partial sums are
//                        combined with fp32 atomics.
//
// Reference ops replaced: Conv2D / Conv2DBackpropInput / Conv2DBackpropFilter for
// advoc_model.py:91-94 (encoder_1), :153-158 (decoder_1), :185-188 (layer_1), :199-202 (layer_5).
#include <string>

#include "conv_internal.h"

namespace advoc {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == ADVOC_ACT_LRELU02) return fmaxf(0.2f * v, v);
  if (act == ADVOC_ACT_RELU) return fmaxf(v, 0.f);
  return v;
}

__device__ __forceinline__ float act_bwd(float x, int act) {
  if (act == ADVOC_ACT_LRELU02) return x > 0.f ? 1.f : 0.2f;
  if (act == ADVOC_ACT_RELU) return x > 0.f ? 1.f : 0.f;
  return 1.f;
}

// ---------------------------------------------------------------------------------------------
// thin_k_gemm
// ---------------------------------------------------------------------------------------------
// KP: padded contraction length (16 or 32); NT: 32-channel tiles per wave.
template <int KP, int NT, bool B_KN>
__global__ __launch_bounds__(256) void thin_k_gemm_kernel(const GatherGemmParams p, int n_base_tiles) {
  __shared__ int s_pix[4][2][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l32 = lane & 31;
  const int phase = blockIdx.z;
  const int ktot = p.c0 + p.c1;            // 1 or 2
  const int kreal = p.ntaps * ktot;
  const int N = p.n_total;
  const int n0 = blockIdx.y * (32 * NT);
  const int64_t M = (int64_t)p.batch * p.gh * p.gw;

  // ---- per-lane K slots: slot s of this lane is k' = 2 s + half ----
  int k_dy[KP / 2], k_dx[KP / 2];
  unsigned k_valid = 0, k_second = 0;
  float wreg[NT][KP / 2];
#pragma unroll
  for (int s = 0; s < KP / 2; ++s) {
    const int kk = 2 * s + half;
    const bool ok = kk < kreal;
    const int t = ok ? kk / ktot : 0, ci = ok ? kk % ktot : 0;
    const int tp = p.tap[phase][t];
    k_dy[s] = (int)(int8_t)(tp & 0xff);
    k_dx[s] = (int)(int8_t)((tp >> 8) & 0xff);
    const int wtap = tp >> 16;
    if (ok) k_valid |= 1u << s;
    if (ok && ci >= p.c0) k_second |= 1u << s;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + 32 * j + l32;
      float w = 0.f;
      if (ok && n < N)
        w = B_KN ? p.w[((int64_t)wtap * ktot + ci) * N + n] : p.w[((int64_t)wtap * N + n) * ktot + ci];
      wreg[j][s] = w;
    }
  }
  float bias[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + 32 * j + l32;
    bias[j] = (p.bias && n < N) ? p.bias[n] : 0.f;
  }

  const int64_t tiles = (M + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t m = tile * 32 + l32;
    const bool live = m < M;
    int img = 0, gy = 0, gx = 0;
    if (live) {
      gx = (int)(m % p.gw);
      const int64_t t = m / p.gw;
      gy = (int)(t % p.gh);
      img = (int)(t / p.gh);
    }
    // output pixel offsets of this wave's 32 rows (lanes 0-31 publish them)
    if (half == 0) {
      int pix0 = -1, pix1 = -1;
      const int oy = gy * p.osy + p.ooy[phase], ox = gx * p.osx + p.oox[phase];
      if (live && oy < p.out_h && ox < p.out_w) {
        pix0 = (img * p.out_h + oy) * p.d[0].pitch + ox;
        pix1 = (img * p.out_h + oy) * p.d[1].pitch + ox;
      }
      s_pix[wave][0][l32] = pix0;
      s_pix[wave][1][l32] = pix1;
    }
    // A operand: one gathered scalar per K slot
    float a[KP / 2];
#pragma unroll
    for (int s = 0; s < KP / 2; ++s) {
      float v = 0.f;
      const int iy = gy * p.sy + k_dy[s], ix = gx * p.sx + k_dx[s];
      if (live && ((k_valid >> s) & 1u) && (unsigned)iy < (unsigned)p.in_h && (unsigned)ix < (unsigned)p.in_w) {
        const bool second = (k_second >> s) & 1u;
        const float* src = second ? p.a1 : p.a0;
        const int cs = second ? p.c1 : p.c0;
        const int pitch = second ? p.a1_pitch : p.a0_pitch;
        // with <= 2 channels per source the channel offset inside a source is k' % ktot - c0 or 0
        const int kk = 2 * s + half;
        const int ci = kk % ktot;
        const int64_t off = (((int64_t)img * p.a_h + iy) * pitch + ix) * cs + (second ? ci - p.c0 : ci);
        v = src[off];
        if (p.in_scale) v = v * p.in_scale[ci] + p.in_shift[ci];
        v = act_fwd(v, p.in_act);
        if (p.a_mask) v *= p.a_mask[off] * p.a_mask_scale;
      }
      a[s] = v;
    }
    floatx16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int s = 0; s < KP / 2; ++s)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wreg[j][s], acc[j], 0, 0, 0);

    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + 32 * j + l32;
      if (n >= N) continue;
      const int di = n >= p.n_split ? 1 : 0;
      const GemmDest& d = p.d[di];
      if (d.p == nullptr) continue;
      const int ch = di ? n - p.n_split : n;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int pix = s_pix[wave][di][row];
        if (pix < 0) continue;
        const int64_t off = (int64_t)pix * d.c + ch;
        float v = acc[j][r] + bias[j];
        if (p.y_mask) v *= p.y_mask[off] * p.y_mask_scale;
        if (p.grad_act != ADVOC_ACT_NONE) v *= act_bwd(d.xpre[off], p.grad_act);
        if (d.accum) v += d.p[off];
        d.p[off] = v;
      }
    }
    wave_lds_sync();
  }
}

template <int KP, bool B_KN>
int launch_thin_k(const GatherGemmParams& p, hipStream_t stream, const char** name_only) {
  const int N = p.n_total;
  const int nt = N % 128 == 0 ? 4 : (N % 64 == 0 ? 2 : 1);
  if (name_only) {
    static std::string names[3];
    const int i = nt == 4 ? 2 : (nt == 2 ? 1 : 0);
    if (names[i].empty())
      names[i] = std::string("thin_k_gemm_kernel<") + std::to_string(KP) + ", " + std::to_string(nt) + ", " +
                 (B_KN ? "true" : "false") + ">";
    *name_only = names[i].c_str();
    return ADVOC_OK;
  }
  const int64_t M = (int64_t)p.batch * p.gh * p.gw;
  const int64_t tiles = ceil_div(M, 32);
  int64_t bx = ceil_div(tiles, 4);
  const int by = (N + 32 * nt - 1) / (32 * nt);
  const int64_t cap = 2048 / (by * p.nphase) > 0 ? 2048 / (by * p.nphase) : 1;
  if (bx > cap) bx = cap;       // grid-stride over row tiles: weights stay in registers
  dim3 grid((unsigned)bx, (unsigned)by, (unsigned)p.nphase);
  ADVOC_CLEAR_LAUNCH_ERROR();
  if (nt == 4) hipLaunchKernelGGL((thin_k_gemm_kernel<KP, 4, B_KN>), grid, dim3(256), 0, stream, p, 0);
  else if (nt == 2) hipLaunchKernelGGL((thin_k_gemm_kernel<KP, 2, B_KN>), grid, dim3(256), 0, stream, p, 0);
  else hipLaunchKernelGGL((thin_k_gemm_kernel<KP, 1, B_KN>), grid, dim3(256), 0, stream, p, 0);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

// ---------------------------------------------------------------------------------------------
// thin_wgrad
// ---------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const WgradParams p, int chunk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l32 = lane & 31;
  const int ca = p.P.c0 + p.P.c1;            // 1 or 2
  const int cb = p.Q.c0 + p.Q.c1;
  const int rows = p.ntaps * ca;             // <= 32 live rows of the A operand
  const int b0 = blockIdx.y * (32 * NT);
  const int64_t M = (int64_t)p.batch * p.gh * p.gw;

  // this lane's A row: (tap, a)
  const bool row_ok = l32 < rows;
  const int t = row_ok ? l32 / ca : 0, a = row_ok ? l32 % ca : 0;
  const int tp = p.tap[t];
  const int dy = (int)(int8_t)(tp & 0xff), dx = (int)(int8_t)((tp >> 8) & 0xff), wtap = tp >> 16;
  const bool a_second = a >= p.P.c0;
  const float* psrc = a_second ? p.P.p1 : p.P.p0;
  const int pcs = a_second ? p.P.c1 : p.P.c0;
  const int ppitch = a_second ? p.P.pitch1 : p.P.pitch0;
  const int pch = a_second ? a - p.P.c0 : a;

  // each wave owns a contiguous run of grid points, two per MFMA (lanes 0-31 / 32-63)
  const int64_t w_begin = ((int64_t)blockIdx.x * 4 + wave) * chunk;
  const int64_t w_end = w_begin + chunk < M ? w_begin + chunk : M;
  int64_t g = w_begin + half;
  int gx = 0, gy = 0, img = 0;
  if (g < M) {
    gx = (int)(g % p.gw);
    const int64_t tt = g / p.gw;
    gy = (int)(tt % p.gh);
    img = (int)(tt / p.gh);
  }

  floatx16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  for (; g - half < w_end; g += 2) {
    const bool live = g < w_end;
    float av = 0.f;
    float bv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bv[j] = 0.f;
    if (live) {
      const int y = gy * p.sy + dy, x = gx * p.sx + dx;
      if (row_ok && (unsigned)y < (unsigned)p.P.h && (unsigned)x < (unsigned)p.P.w) {
        const int64_t off = (((int64_t)img * p.P.h + y) * ppitch + x) * pcs + pch;
        float v = psrc[off];
        if (p.P.scale) v = v * p.P.scale[a] + p.P.shift[a];
        v = act_fwd(v, p.P.act);
        if (p.P.mask) v *= p.P.mask[off] * p.P.mask_scale;
        av = v;
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int b = b0 + 32 * j + l32;
        if (b < cb) {
          const bool second = b >= p.Q.c0;
          const float* src = second ? p.Q.p1 : p.Q.p0;
          const int cs = second ? p.Q.c1 : p.Q.c0;
          const int pitch = second ? p.Q.pitch1 : p.Q.pitch0;
          const int64_t off = (((int64_t)img * p.Q.h + gy) * pitch + gx) * cs + (second ? b - p.Q.c0 : b);
          float v = src[off];
          if (p.Q.scale) v = v * p.Q.scale[b] + p.Q.shift[b];
          v = act_fwd(v, p.Q.act);
          if (p.Q.mask) v *= p.Q.mask[off] * p.Q.mask_scale;
          bv[j] = v;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[j], acc[j], 0, 0, 0);
    // advance this lane's grid point by 2
    gx += 2;
    while (gx >= p.gw) {
      gx -= p.gw;
      if (++gy >= p.gh) { gy = 0; ++img; }
    }
  }

  // acc[j][r]: row (r&3)+8(r>>2)+4*half = A row index (tap, a); column l32 = channel
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int b = b0 + 32 * j + l32;
    if (b >= cb) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      if (row >= rows) continue;
      const int rt = row / ca, ra = row % ca;
      const int rw = p.tap[rt] >> 16;
      unsafeAtomicAdd(p.dw + ((int64_t)rw * ca + ra) * cb + b, acc[j][r]);
    }
  }
  (void)wtap;
}

}  // namespace

int launch_thin_k_gemm(const GatherGemmParams& p, bool b_kn, hipStream_t stream, const char** name_only) {
  const int ktot = p.c0 + p.c1;
  if (ktot < 1 || ktot > 2 || p.c0 > 2 || p.n_total % 32) return ADVOC_ERR_UNSUPPORTED;
  if (p.a_mask && p.c1) return ADVOC_ERR_UNSUPPORTED;
  const int kreal = p.ntaps * ktot;
  if (kreal > 32) return ADVOC_ERR_UNSUPPORTED;
  if ((int64_t)p.batch * p.out_h * (int64_t)(p.d[0].pitch > p.d[1].pitch ? p.d[0].pitch : p.d[1].pitch) > 0x7fffffffLL)
    return ADVOC_ERR_UNSUPPORTED;
  if (kreal <= 16) return b_kn ? launch_thin_k<16, true>(p, stream, name_only) : launch_thin_k<16, false>(p, stream, name_only);
  return b_kn ? launch_thin_k<32, true>(p, stream, name_only) : launch_thin_k<32, false>(p, stream, name_only);
}

int launch_wgrad_thin_mfma(const WgradParams& p, hipStream_t stream, const char** name_only) {
  const int ca = p.P.c0 + p.P.c1, cb = p.Q.c0 + p.Q.c1;
  if (ca < 1 || ca > 2 || p.ntaps * ca > 32 || cb % 32) return ADVOC_ERR_UNSUPPORTED;
  const int nt = cb % 128 == 0 ? 4 : (cb % 64 == 0 ? 2 : 1);
  if (name_only) {
    *name_only = nt == 4 ? "thin_wgrad_kernel<4>" : (nt == 2 ? "thin_wgrad_kernel<2>" : "thin_wgrad_kernel<1>");
    return ADVOC_OK;
  }
  hipError_t e = hipMemsetAsync(p.dw, 0, sizeof(float) * (size_t)p.ntaps * ca * cb, stream);
  if (e != hipSuccess) { note_hip_error(e); return ADVOC_ERR_HIP; }
  const int64_t M = (int64_t)p.batch * p.gh * p.gw;
  const int by = cb / (32 * nt);
  // ~2048 waves over the pixel axis (8 per CU) so HBM latency is covered by occupancy
  int64_t waves = 2048 / by;
  if (waves < 4) waves = 4;
  int64_t chunk = ceil_div(M, waves);
  chunk = (chunk + 1) / 2 * 2;
  if (chunk < 64) chunk = 64;
  const int64_t bx = ceil_div(ceil_div(M, chunk), 4);
  if (chunk > 0x7fffffffLL || bx > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
  dim3 grid((unsigned)bx, (unsigned)by);
  ADVOC_CLEAR_LAUNCH_ERROR();
  if (nt == 4) hipLaunchKernelGGL(thin_wgrad_kernel<4>, grid, dim3(256), 0, stream, p, (int)chunk);
  else if (nt == 2) hipLaunchKernelGGL(thin_wgrad_kernel<2>, grid, dim3(256), 0, stream, p, (int)chunk);
  else hipLaunchKernelGGL(thin_wgrad_kernel<1>, grid, dim3(256), 0, stream, p, (int)chunk);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

}  // namespace advoc
