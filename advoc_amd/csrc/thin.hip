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
#include "image_emit.h"
#include "tuning.h"
#include "x6.h"

namespace advoc {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// buffer offset that is out of range for every descriptor of the epilogues below (destinations are < 4 GB - 256 B:
// checked by the launchers)
constexpr unsigned kThinOob = 0xffffff00u;

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
// Per-slot gather constants and the weight tile live in LDS (read back with conflict-free
// ds_read_b32), so a wave needs ~16 NT accumulator registers + KP/2 operands: 4-8 waves per SIMD
// cover the HBM latency of this streaming kernel.
constexpr int kThinPatchRows = 4;                       // tap rows a tile can reach
constexpr int kThinPatchCols = 31 * 2 + 4;              // 32 grid points at stride <= 2 + tap span
constexpr int kThinPatch = kThinPatchRows * kThinPatchCols * 2;   // x up to 2 channels

// PL: patch elements per lane (ceil(pr * pc * ktot / 64)); a template parameter because the fetch
// registers (value + mask factor + coordinates per element) decide the occupancy of the wide-N cases
// FW (r4): -1 = everything decided at run time (masks, accumulation, images: the r2 kernel); -2 = a BACKWARD-DATA call that only
// gates on the pre-activation values (no masks, no accumulation, no image consumers: what the models' calls are) -- the generic
// instance executed the image arithmetic of two absent consumers for every value, half its epilogue; 0 / 1 / 2 = a FORWARD
// call without masks / gating / accumulation and with that many image consumers in oimg[0 .. FW): the epilogue loads (12
// buffer instructions per 32-channel block, out of range and dropped -- but issued) are not there and the image arithmetic
// of a consumer that does not exist is not executed; -3 = as -2 plus ONE image: the output-gradient image of the layer below
// and that tensor's per-channel sums, its bias gradient (advoc_conv_layer.dx_img; 176 registers, two waves per SIMD -- at
// three, 8 spilled, it is slower: profiles/r04_i_emit_dx_ab.txt).  The kernel is bound by what it ISSUES (rocprofv3: issuing 0.98 at three
// waves per SIMD, ~1 500 vector instructions per tile; an unused consumer's share is 15 % of them).
template <int KP, int NT, bool B_KN, int PL, int FW>
__global__ __launch_bounds__(256, NT == 4 ? 2 : 3) void thin_k_gemm_kernel(const GatherGemmParams p, int dy_min, int dx_min,
                                                          int pr, int pc, int tiles_x) {
  __shared__ int s_pix[4][2][32];
  __shared__ int s_delta[KP];     // LDS-patch offset of slot k' relative to the lane's column origin
  __shared__ int s_info[KP];      // valid << 16 | ci << 18
  __shared__ float s_w[KP][32 * NT];
  __shared__ __attribute__((aligned(16))) float s_T[4][32 * 36];
  // The 1-2 channel input a tile of 32 grid points (one grid row) reads: pr tap rows x pc columns
  // x ktot channels, staged per wave with coalesced loads (zeros outside the image, affine /
  // activation / mask applied once here), so the 16-32 operand gathers per lane are ds_reads at
  // tile-invariant offsets.  Gathering them from global memory made this kernel latency-bound
  // (213 -> 88 us on discriminator layer_1 with the loads stubbed out).
  __shared__ float s_patch[4][kThinPatch];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // uniform for the compiler too: the tile index and everything derived from it (image, row, origins) is scalar arithmetic
  const int half = lane >> 5, l32 = lane & 31;
  const int phase = blockIdx.z;
  const int ktot = p.c0 + p.c1;            // 1 or 2
  const int kreal = p.ntaps * ktot;
  const int N = p.n_total;
  const int n0 = blockIdx.y * (32 * NT);
  const float slope = p.in_act == ADVOC_ACT_LRELU02 ? 0.2f : (p.in_act == ADVOC_ACT_RELU ? 0.f : 1.f);
  const float gslope = p.grad_act == ADVOC_ACT_LRELU02 ? 0.2f : (p.grad_act == ADVOC_ACT_RELU ? 0.f : 1.f);

  for (int kk = threadIdx.x; kk < KP; kk += 256) {
    const bool ok = kk < kreal;
    const int t = ok ? kk / ktot : 0, ci = ok ? kk % ktot : 0;
    const int tp = p.tap[phase][t];
    const int dy = (int)(int8_t)(tp & 0xff), dx = (int)(int8_t)((tp >> 8) & 0xff);
    s_delta[kk] = ((dy - dy_min) * pc + (dx - dx_min)) * ktot + ci;
    s_info[kk] = ((ok ? 1 : 0) << 16) | (ci << 18);
  }
  for (int idx = threadIdx.x; idx < KP * 32 * NT; idx += 256) {
    const int kk = idx / (32 * NT), nn = idx % (32 * NT);
    const int n = n0 + nn;
    float w = 0.f;
    if (kk < kreal && n < N) {
      const int t = kk / ktot, ci = kk % ktot;
      const int wtap = p.tap[phase][t] >> 16;
      w = B_KN ? p.w[((int64_t)wtap * ktot + ci) * N + n] : p.w[((int64_t)wtap * N + n) * ktot + ci];
    }
    s_w[kk][nn] = w;
  }
  float bias[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + 32 * j + l32;
    bias[j] = (p.bias && n < N) ? p.bias[n] : 0.f;
  }
  // patch elements this lane stages every tile: element e = lane + 64 i -> (row, col, channel)
  const int patch_elems = pr * pc * ktot;
  int pe[PL];                       // row | col << 3 | channel << 12, -1 past the patch (one register per element)
  // (decoded where used, from a copy the compiler cannot see through: hoisted out of the tile loop the three fields
  // would live in three registers per element again)
#define PE_ROW(i) (pq_##i < 0 ? -1 : (pq_##i & 7))
#define PE_COL(i) ((pq_##i >> 3) & 0x1ff)
#define PE_CI(i) ((pq_##i >> 12) & 1)
#define PE_OPEN(i) int pq_##i = pe[i]; asm volatile("" : "+v"(pq_##i));
#pragma unroll
  for (int i = 0; i < PL; ++i) {
    const int e = lane + 64 * i;
    const int ee = e < patch_elems ? e : 0;
    const int ci_ = ee % ktot;
    const int px = ee / ktot;
    pe[i] = e < patch_elems ? ((px / pc) | ((px % pc) << 3) | (ci_ << 12)) : -1;
  }
  __syncthreads();
  // this lane's K slots (k' = 2 s + half): LDS offsets relative to the tile's patch, decoded once
  int sl_off[KP / 2];
#pragma unroll
  for (int s = 0; s < KP / 2; ++s) {
    const int kk = 2 * s + half;
    sl_off[s] = ((s_info[kk] >> 16) & 1) ? s_delta[kk] + l32 * p.sx * ktot : -1;
  }
  float* patch = &s_patch[wave][0];

  // tile = 32 consecutive grid points of ONE grid row (32-bit arithmetic: launch checks the range)
  const int tiles = p.batch * p.gh * tiles_x;
  float pv[PL], pm[PL];
  // loads the patch of `TILE` into registers (the NEXT tile's loads fly during this tile's MFMAs
  // and stores: a wave has no other way to overlap its own memory latency)
#define ADVOC_THIN_FETCH(TILE)                                                                        \
  {                                                                                                   \
    const int rowid_ = (TILE) / tiles_x;                                                              \
    const int gx0_ = ((TILE) - rowid_ * tiles_x) * 32;                                                \
    const int img_ = rowid_ / p.gh, gy_ = rowid_ - img_ * p.gh;                                       \
    const int iy0_ = gy_ * p.sy + dy_min, ix0_ = gx0_ * p.sx + dx_min;                                \
    _Pragma("unroll") for (int i = 0; i < PL; ++i) {                                     \
      PE_OPEN(i)                                                                                      \
      const int iy = iy0_ + PE_ROW(i), ix = ix0_ + PE_COL(i);                                         \
      float v = 0.f, mk = 1.f;                                                                        \
      if (PE_ROW(i) >= 0 && (unsigned)iy < (unsigned)p.in_h && (unsigned)ix < (unsigned)p.in_w) {     \
        const bool second = PE_CI(i) >= p.c0;                                                         \
        const int off = second ? ((img_ * p.a_h + iy) * p.a1_pitch + ix) * p.c1 + (PE_CI(i) - p.c0)   \
                               : ((img_ * p.a_h + iy) * p.a0_pitch + ix) * p.c0 + PE_CI(i);           \
        v = (second ? p.a1 : p.a0)[off];                                                              \
        if (p.a_mask && !second) mk = p.a_mask[off] * p.a_mask_scale;                                 \
      }                                                                                               \
      pv[i] = v;                                                                                      \
      pm[i] = mk;                                                                                     \
    }                                                                                                 \
  }
  // what the epilogue of one 32-channel column block loads (see there)
  struct EpiLoads {
    unsigned off[4];               // element offset of the row's 4 channels in its destination, kThinOob without a pixel
    u32x4 xp[4];
    unsigned gm[4], ym[4];
  };
  EpiLoads epi;
#define ADVOC_THIN_PRELOAD(J)                                                                                     \
  {                                                                                                               \
    const int nt0_ = n0 + 32 * (J);                                                                               \
    const int di_ = nt0_ >= p.n_split ? 1 : 0;                                                                    \
    const GemmDest& d_ = p.d[di_];                                                                                \
    const bool ok_ = nt0_ < N && d_.p != nullptr;                                                                 \
    const int ch_ = (di_ ? nt0_ - p.n_split : nt0_) + 4 * (lane & 7);                                             \
    const bool grad_ = ok_ && p.grad_act != ADVOC_ACT_NONE;                                                       \
    const __amdgpu_buffer_rsrc_t rs_x_ = __builtin_amdgcn_make_buffer_rsrc(                                      \
        const_cast<float*>(grad_ ? d_.xpre : d_.p), 0, grad_ ? kThinOob : 0u, 0x00020000);                        \
    const __amdgpu_buffer_rsrc_t rs_g_ = __builtin_amdgcn_make_buffer_rsrc(                                      \
        (ok_ && d_.gmask) ? const_cast<uint8_t*>(d_.gmask) : reinterpret_cast<uint8_t*>(d_.p), 0,                 \
        (ok_ && d_.gmask) ? kThinOob : 0u, 0x00020000);                                                           \
    const __amdgpu_buffer_rsrc_t rs_y_ = __builtin_amdgcn_make_buffer_rsrc(                                      \
        (ok_ && p.y_mask) ? const_cast<uint8_t*>(p.y_mask) : reinterpret_cast<uint8_t*>(d_.p), 0,                 \
        (ok_ && p.y_mask) ? kThinOob : 0u, 0x00020000);                                                           \
    EpiLoads& e_ = epi;                                                                                  \
    _Pragma("unroll") for (int ps = 0; ps < 4; ++ps) {                                                            \
      const int pix = s_pix[wave][di_][(lane >> 3) + 8 * ps];                                                     \
      e_.off[ps] = pix < 0 ? kThinOob : (unsigned)(pix * d_.c + ch_);                                             \
      if (FW < 0) {                                                                                               \
        const unsigned ob = pix < 0 ? kThinOob : e_.off[ps] * 4u;                                                 \
        e_.xp[ps] = __builtin_amdgcn_raw_buffer_load_b128(rs_x_, ob, 0, 0);                                       \
        if (FW == -1) {                                                                                           \
          e_.gm[ps] = __builtin_amdgcn_raw_buffer_load_b32(rs_g_, e_.off[ps], 0, 0);                              \
          e_.ym[ps] = __builtin_amdgcn_raw_buffer_load_b32(rs_y_, e_.off[ps], 0, 0);                              \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
  }
  // the bias was loaded before the loop: consume it once here, so that its use inside the loop is not a wait for "every
  // memory operation in flight" (the compiler cannot tell loop iterations apart) -- that wait included the stores
#pragma unroll
  for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(bias[j]));
  const int tile0 = blockIdx.x * 4 + wave;
  // forward launches: the consumers' operand images of the output (image_emit.h)
  const ImgOut o0 = p.oimg[0], o1 = p.oimg[1];
  const bool emit0 = o0.img != nullptr, emit1 = o1.img != nullptr;
  // (r5, FW == -3 with GatherGemmParams::oimg_bounded) the a-priori scale of igemm_patch.hip: |dx| <= max|dy| max|w| taps K,
  // here with max|dy| in *a_amax (a magnitude pass over the thin operand in front of the launch) and max|w| in *w_amax
  float eup0_ = 1.f;
  if (emit0) {
    if (FW == -3 && p.oimg_bounded) {
      eup0_ = emit_up_scale_bounded(__uint_as_float(*p.a_amax) * __uint_as_float(*p.w_amax) * (float)(p.ntaps * (p.c0 + p.c1)));
    } else if (FW >= 1 && p.oimg_bounded) {
      // forward: |y| <= max|x| max|w| taps K + max|b| (the input activation has slope <= 1)
      float bm = 0.f;
      if (p.bias)
        for (int n = lane; n < N; n += 64) bm = fmaxf(bm, fabsf(p.bias[n]));
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) bm = fmaxf(bm, __shfl_xor(bm, off, 64));
      eup0_ = emit_up_scale_bounded(__uint_as_float(*p.a_amax) * __uint_as_float(*p.w_amax) * (float)(p.ntaps * (p.c0 + p.c1)) + bm);
    } else {
      eup0_ = emit_up_scale(o0.hdr[2]);
    }
  }
  const float eup0 = eup0_, eup1 = emit1 ? emit_up_scale(o1.hdr[2]) : 1.f;
  float evmax0 = 0.f, evmax1 = 0.f;
  float csum[FW == -3 ? NT : 1][4] = {};
  if (threadIdx.x == 0) {
    if (emit0) o0.hdr[1] = __float_as_uint(1.f / eup0);
    if (emit1) o1.hdr[1] = __float_as_uint(1.f / eup1);
  }
  const __amdgpu_buffer_rsrc_t rs_e0 = __builtin_amdgcn_make_buffer_rsrc(
      emit0 ? reinterpret_cast<void*>(o0.img) : reinterpret_cast<void*>(p.d[0].p), 0, emit0 ? kThinOob : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_e1 = __builtin_amdgcn_make_buffer_rsrc(
      emit1 ? reinterpret_cast<void*>(o1.img) : reinterpret_cast<void*>(p.d[0].p), 0, emit1 ? kThinOob : 0u, 0x00020000);
  if (tile0 < tiles) ADVOC_THIN_FETCH(tile0);
  for (int tile = tile0; tile < tiles; tile += gridDim.x * 4) {
    const int rowid = tile / tiles_x;
    const int gx0 = (tile - rowid * tiles_x) * 32;
    const int img = rowid / p.gh, gy = rowid - img * p.gh;
    const int gx = gx0 + l32;
    const bool live = gx < p.gw;
    if (half == 0) {   // lanes 0-31 publish their row's output pixel
      int pix0 = -1, pix1 = -1;
      const int oy = gy * p.osy + p.ooy[phase], ox = gx * p.osx + p.oox[phase];
      if (live && oy < p.out_h && ox < p.out_w) {
        pix0 = (img * p.out_h + oy) * p.d[0].pitch + ox;
        pix1 = (img * p.out_h + oy) * p.d[1].pitch + ox;
      }
      s_pix[wave][0][l32] = pix0;
      s_pix[wave][1][l32] = pix1;
    }
    // ---- park the fetched patch in LDS (affine + activation here; zero padding stays zero) ----
    {
      const int iy0 = gy * p.sy + dy_min, ix0 = gx0 * p.sx + dx_min;
#pragma unroll
      for (int i = 0; i < PL; ++i) {
        PE_OPEN(i)
        if (PE_ROW(i) < 0) continue;
        float v = pv[i];
        const int iy = iy0 + PE_ROW(i), ix = ix0 + PE_COL(i);
        const bool inb = (unsigned)iy < (unsigned)p.in_h && (unsigned)ix < (unsigned)p.in_w;
        if (p.in_scale) v = inb ? v * p.in_scale[PE_CI(i)] + p.in_shift[PE_CI(i)] : 0.f;
        v = fmaxf(v, slope * v);
        patch[lane + 64 * i] = v * pm[i];
      }
    }
    wave_lds_sync();
    {
      const int next = tile + gridDim.x * 4;
      if (next < tiles) ADVOC_THIN_FETCH(next);
    }
    ADVOC_THIN_PRELOAD(0)
    // A operand: one scalar per K slot from the patch
    float a[KP / 2];
#pragma unroll
    for (int s = 0; s < KP / 2; ++s) a[s] = (live && sl_off[s] >= 0) ? patch[sl_off[s]] : 0.f;
    floatx16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int s = 0; s < KP / 2; ++s)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], s_w[2 * s + half][32 * j + l32], acc[j], 0, 0, 0);

    wave_lds_sync();
    // ---- epilogue: each 32x32 tile is transposed through a private LDS patch (16-byte accesses on 128-byte rows) ----
    // Global loads and stores retire through ONE in-order counter, so `load x; wait; store` per row pays a memory round
    // trip per KILOBYTE (measured: decoder_1 backward-data at 1.4 TB/s).  All accesses are buffer instructions whose
    // offset is out of range for rows without a pixel and for tensors this launch does not have (loads return zero,
    // stores are dropped): no branch around any of them.  A column block issues all its loads first (block 0 before the
    // MFMAs), transposes while they fly, then stores: one round trip per 32-channel block instead of one per row.
    float* T = &s_T[wave][0];
    const int trow = lane >> 3, tq = lane & 7;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (j > 0) ADVOC_THIN_PRELOAD(j)              // (block 0's loads went out before the MFMAs)
      EpiLoads& e = epi;
#pragma unroll
      for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * half) * 36 + l32] = acc[j][r] + bias[j];
      wave_lds_sync();
      float4 v[4];
      unsigned so[4];
      const int di = (n0 + 32 * j) >= p.n_split ? 1 : 0;
      const GemmDest& d = p.d[di];
      const bool okj = n0 + 32 * j < N && d.p != nullptr;
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        v[ps] = *reinterpret_cast<const float4*>(T + (trow + 8 * ps) * 36 + 4 * tq);
        if (FW == -1 && p.y_mask) {
          const unsigned mk = e.ym[ps];
          v[ps].x *= (float)(mk & 0xffu) * p.y_mask_scale; v[ps].y *= (float)((mk >> 8) & 0xffu) * p.y_mask_scale;
          v[ps].z *= (float)((mk >> 16) & 0xffu) * p.y_mask_scale; v[ps].w *= (float)(mk >> 24) * p.y_mask_scale;
        }
        if (FW < 0 && p.grad_act != ADVOC_ACT_NONE) {
          float4 x = make_float4(__uint_as_float(e.xp[ps].x), __uint_as_float(e.xp[ps].y), __uint_as_float(e.xp[ps].z),
                                 __uint_as_float(e.xp[ps].w));
          if (d.gscale) {
            const int ch = (di ? n0 + 32 * j - p.n_split : n0 + 32 * j) + 4 * tq;
            const float4 gs = *reinterpret_cast<const float4*>(d.gscale + ch);
            const float4 gh = *reinterpret_cast<const float4*>(d.gshift + ch);
            x.x = x.x * gs.x + gh.x; x.y = x.y * gs.y + gh.y; x.z = x.z * gs.z + gh.z; x.w = x.w * gs.w + gh.w;
          }
          v[ps].x *= x.x > 0.f ? 1.f : gslope; v[ps].y *= x.y > 0.f ? 1.f : gslope;
          v[ps].z *= x.z > 0.f ? 1.f : gslope; v[ps].w *= x.w > 0.f ? 1.f : gslope;
        }
        if (FW == -1 && d.gmask) {
          const unsigned mk = e.gm[ps];
          v[ps].x *= (float)(mk & 0xffu) * d.gmask_scale; v[ps].y *= (float)((mk >> 8) & 0xffu) * d.gmask_scale;
          v[ps].z *= (float)((mk >> 16) & 0xffu) * d.gmask_scale; v[ps].w *= (float)(mk >> 24) * d.gmask_scale;
        }
        so[ps] = e.off[ps] == kThinOob ? kThinOob : e.off[ps] * 4u;
      }
      if (FW == -1 && okj && d.accum) {          // (no thin layer of the models accumulates: not worth 16 registers of prefetch)
        const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(d.p, 0, kThinOob, 0x00020000);
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const u32x4 o = __builtin_amdgcn_raw_buffer_load_b128(rs_o, so[ps], 0, 0);
          v[ps].x += __uint_as_float(o.x); v[ps].y += __uint_as_float(o.y);
          v[ps].z += __uint_as_float(o.z); v[ps].w += __uint_as_float(o.w);
        }
      }
      // (FW == -3, d0_no_store: destination 0 exists as the image only)
      const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(
          d.p, 0, (okj && !((FW == -3 || FW >= 1) && p.d0_no_store && di == 0)) ? kThinOob : 0u, 0x00020000);
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        u32x4 sv;
        sv.x = __float_as_uint(v[ps].x); sv.y = __float_as_uint(v[ps].y);
        sv.z = __float_as_uint(v[ps].z); sv.w = __float_as_uint(v[ps].w);
        __builtin_amdgcn_raw_buffer_store_b128(sv, rs_d, so[ps], 0, 0);
      }
      // consumers' operand images of the output (forward launches; image_emit.h): unconditional buffer stores, dropped by
      // the range check for rows without a pixel and when there is no such consumer
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const bool ok = okj && di == 0 && so[ps] != kThinOob;
        if (FW == -1 || FW >= 1 || FW == -3) emit4_buffer(rs_e0, o0.slope, eup0, v[ps], ok ? so[ps] : kThinOob, ok && emit0, evmax0);
        if (FW == -1 || FW >= 2) emit4_buffer(rs_e1, o1.slope, eup1, v[ps], ok ? so[ps] : kThinOob, ok && emit1, evmax1);
        if constexpr (FW == -3) {         // column sums of the destination: the lane's four channels over the rows it stores
          const float m = ok ? 1.f : 0.f;
          csum[j][0] = fmaf(m, v[ps].x, csum[j][0]); csum[j][1] = fmaf(m, v[ps].y, csum[j][1]);
          csum[j][2] = fmaf(m, v[ps].z, csum[j][2]); csum[j][3] = fmaf(m, v[ps].w, csum[j][3]);
        }
      }
      wave_lds_sync();
    }
  }
#undef ADVOC_THIN_PRELOAD
  if (emit0) emit_finish(o0, eup0, evmax0);
  if (emit1) emit_finish(o1, eup1, evmax1);
  if constexpr (FW == -3) if (p.ocolsum_table) {
    // lanes (row lane >> 3, channel quad lane & 7): fold the eight rows, then one atomic per channel and wave into one of the
    // replica tables (image.hip: thousands of waves adding to the same N addresses serialise in the L2)
    // (the table's rows are the channels of destination 0: a second destination -- the skip source of a decoder -- has no sums)
    const int cs_c = p.d[0].c;
    float* tab = p.ocolsum_table + (size_t)((blockIdx.x * 4 + wave) & (kColsumReplicas - 1)) * cs_c;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float t = csum[j][c];
        t += __shfl_xor(t, 8, 64);
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        const int n = n0 + 32 * j + 4 * (lane & 7) + c;
        if (lane < 8 && n < N && n < p.n_split && n < cs_c) unsafeAtomicAdd(tab + n, t);
      }
  }
}

#undef PE_ROW
#undef PE_COL
#undef PE_CI
#undef PE_OPEN

template <int KP, bool B_KN>
int launch_thin_k(const GatherGemmParams& p, hipStream_t stream, const char** name_only) {
  const int N = p.n_total;
  const int nt = N % 128 == 0 ? 4 : (N % 64 == 0 ? 2 : 1);
  int dy_min = 127, dy_max = -128, dx_min = 127, dx_max = -128;
  for (int ph = 0; ph < p.nphase; ++ph)
    for (int t = 0; t < p.ntaps; ++t) {
      const int dy = (int)(int8_t)(p.tap[ph][t] & 0xff), dx = (int)(int8_t)((p.tap[ph][t] >> 8) & 0xff);
      dy_min = dy < dy_min ? dy : dy_min; dy_max = dy > dy_max ? dy : dy_max;
      dx_min = dx < dx_min ? dx : dx_min; dx_max = dx > dx_max ? dx : dx_max;
    }
  const int pr = dy_max - dy_min + 1, pc = 31 * p.sx + (dx_max - dx_min) + 1;
  if (pr > kThinPatchRows || pc > kThinPatchCols) return ADVOC_ERR_UNSUPPORTED;
  const int tiles_x = (p.gw + 31) / 32;
  const int64_t tiles = (int64_t)p.batch * p.gh * tiles_x;
  if (tiles > 0x7fffffffLL / 64) return ADVOC_ERR_UNSUPPORTED;
  // the epilogue addresses its destinations with 32-bit byte offsets
  for (int di = 0; di < 2; ++di)
    if (p.d[di].p && (int64_t)p.batch * p.out_h * p.d[di].pitch * p.d[di].c * 4 >= (int64_t)kThinOob)
      return ADVOC_ERR_UNSUPPORTED;
  if (name_only) {
    static std::string names[3];
    const int i = nt == 4 ? 2 : (nt == 2 ? 1 : 0);
    if (names[i].empty())
      names[i] = std::string("thin_k_gemm_kernel<") + std::to_string(KP) + ", " + std::to_string(nt) + ", " +
                 (B_KN ? "true" : "false") + ">";
    *name_only = names[i].c_str();
    // the epilogue writes GatherGemmParams.oimg (image_emit.h); 2: as a backward-data call that only gates on the
    // pre-activation values, into ONE image (the layer below's output gradient) with that tensor's column sums on the way
    if (p.emit_report && p.oimg_bounded && p.grad_act == ADVOC_ACT_NONE) {
      // 5: a FORWARD call that writes ONE consumer's image under the a-priori scale (and may leave the fp32 tensor out)
      *p.emit_report = (tuning().thin_fwd_spec && B_KN && nt <= 2 && !p.y_mask && !p.d[0].gmask && !p.d[1].gmask && !p.d[0].accum &&
                        !p.d[1].accum && !p.d[1].p && p.oimg[0].img && !p.oimg[1].img && p.w_amax) ? 5 : 0;
      return ADVOC_OK;
    }
    if (p.emit_report)
      *p.emit_report = (nt >= 2 && p.grad_act != ADVOC_ACT_NONE && !p.y_mask && !p.d[0].gmask && !p.d[1].gmask && !p.d[0].accum &&
                        !p.d[1].accum && (!p.d[1].p || p.oimg_bounded) && p.oimg[0].img && !p.oimg[1].img &&
                        (!p.oimg_bounded || (p.w_amax && p.d[0].c % 32 == 0 && p.n_split % 32 == 0)) && tuning().thin_fwd_spec) ? 2 : 1;
    return ADVOC_OK;
  }
  int64_t bx = ceil_div(tiles, 4);
  const int by = (N + 32 * nt - 1) / (32 * nt);
  // a forward call (nothing for the epilogue to load): the instance compiled for its number of image consumers, which are
  // moved to oimg[0 .. fw)
  GatherGemmParams q = p;
  int fw = -1;
  if (tuning().thin_fwd_spec && p.grad_act == ADVOC_ACT_NONE && !p.y_mask && !p.d[0].gmask && !p.d[1].gmask && !p.d[0].accum &&
      !p.d[1].accum) {
    if (!q.oimg[0].img && q.oimg[1].img) { q.oimg[0] = q.oimg[1]; q.oimg[1].img = nullptr; }
    fw = q.oimg[0].img ? (q.oimg[1].img ? 2 : 1) : 0;
    if (p.oimg_bounded) {
      if (fw != 1 || !B_KN || nt > 2 || !p.w_amax || !p.a_amax || !q.oimg[0].hdr || p.d[1].p) return ADVOC_ERR_UNSUPPORTED;
      hipError_t e = hipMemsetAsync(q.oimg[0].hdr, 0, 4, stream);      // the magnitude accumulator of the image
      if (e != hipSuccess) { note_hip_error(e); return ADVOC_ERR_HIP; }
    }
  } else if (tuning().thin_fwd_spec && p.grad_act != ADVOC_ACT_NONE && !p.y_mask && !p.d[0].gmask && !p.d[1].gmask && !p.d[0].accum &&
             !p.d[1].accum && !p.oimg[0].img && !p.oimg[1].img) {
    fw = -2;
  } else if (tuning().thin_fwd_spec && nt >= 2 && p.grad_act != ADVOC_ACT_NONE && !p.y_mask && !p.d[0].gmask && !p.d[1].gmask &&
             !p.d[0].accum && !p.d[1].accum && (!p.d[1].p || p.oimg_bounded) && p.oimg[0].img && !p.oimg[1].img) {
    fw = -3;        // ... and writes the layer below's output-gradient image (+ its bias column sums)
    if (p.oimg_bounded && (!p.w_amax || !p.a_amax || !p.oimg[0].hdr)) return ADVOC_ERR_NULL;
    if (p.ocolsum_out) {
      if (!p.ocolsum_table) return ADVOC_ERR_NULL;
      hipError_t e = hipMemsetAsync(p.ocolsum_table, 0, sizeof(float) * kColsumReplicas * (size_t)p.d[0].c, stream);
      if (e != hipSuccess) { note_hip_error(e); return ADVOC_ERR_HIP; }
    }
    if (p.oimg_bounded) {      // the magnitude accumulator of the image this launch writes
      hipError_t e = hipMemsetAsync(p.oimg[0].hdr, 0, 4, stream);
      if (e != hipSuccess) { note_hip_error(e); return ADVOC_ERR_HIP; }
    }
  }
  if (fw != -3 && (p.ocolsum_out || (p.grad_act != ADVOC_ACT_NONE && p.oimg[0].img))) return ADVOC_ERR_UNSUPPORTED;
  if (p.oimg_bounded && fw != -3 && fw != 1) return ADVOC_ERR_UNSUPPORTED;
  const int pl = (pr * pc * (p.c0 + p.c1) + 63) / 64;
  dim3 grid(1, (unsigned)by, (unsigned)p.nphase);
  ADVOC_CLEAR_LAUNCH_ERROR();
  // the grid is ONE round of the chip (as many workgroups as are resident at once: occupancy query per instance, cached),
  // grid-striding over the tiles: a second, partial round would leave most CUs idle behind it
#define ADVOC_THIN_LAUNCH(NT_, PL_)                                                                   \
  {                                                                                                   \
    static int per_cu = 0;                                                                            \
    if (per_cu == 0) {                                                                                \
      int nb = 0;                                                                                     \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, thin_k_gemm_kernel<KP, NT_, B_KN, PL_, -1>, 256, 0) != hipSuccess || \
          nb < 1)                                                                                     \
        nb = 2;                                                                                       \
      per_cu = nb;                                                                                    \
    }                                                                                                 \
    const int64_t cap = (int64_t)per_cu * device_cu_count() / (by * p.nphase);                        \
    if (bx > cap) bx = cap > 0 ? cap : 1;                                                             \
    grid.x = (unsigned)bx;                                                                            \
    if (NT_ >= 2 && fw == -2) {             /* the backward-data calls of the models */                      \
      hipLaunchKernelGGL((thin_k_gemm_kernel<KP, NT_, B_KN, PL_, -2>), grid, dim3(256), 0, stream, p, dy_min, dx_min, pr, \
                         pc, tiles_x);                                                                \
    } else if (NT_ >= 2 && fw == -3) {                                                                \
      hipLaunchKernelGGL((thin_k_gemm_kernel<KP, NT_ >= 2 ? NT_ : 2, B_KN, PL_, -3>), grid, dim3(256), 0, stream, p, dy_min, \
                         dx_min, pr, pc, tiles_x);                                                    \
    } else if (B_KN && NT_ <= 2 && fw >= 0) {      /* the forward calls of the models (32 / 64 output channels) */ \
      if (fw == 0) hipLaunchKernelGGL((thin_k_gemm_kernel<KP, NT_ <= 2 ? NT_ : 1, true, PL_, 0>), grid, dim3(256), 0, stream, q, dy_min, dx_min, pr, pc, tiles_x); \
      else if (fw == 1) hipLaunchKernelGGL((thin_k_gemm_kernel<KP, NT_ <= 2 ? NT_ : 1, true, PL_, 1>), grid, dim3(256), 0, stream, q, dy_min, dx_min, pr, pc, tiles_x); \
      else hipLaunchKernelGGL((thin_k_gemm_kernel<KP, NT_ <= 2 ? NT_ : 1, true, PL_, 2>), grid, dim3(256), 0, stream, q, dy_min, dx_min, pr, pc, tiles_x); \
    } else {                                                                                          \
      hipLaunchKernelGGL((thin_k_gemm_kernel<KP, NT_, B_KN, PL_, -1>), grid, dim3(256), 0, stream, p, dy_min, dx_min, pr, \
                         pc, tiles_x);                                                                \
    }                                                                                                 \
  }
#define ADVOC_THIN_LAUNCH_PL(NT_)                                                                     \
  {                                                                                                   \
    if (pl <= 3) ADVOC_THIN_LAUNCH(NT_, 3)                                                            \
    else if (pl <= 5) ADVOC_THIN_LAUNCH(NT_, 5)                                                       \
    else ADVOC_THIN_LAUNCH(NT_, 9)                                                                    \
  }
  if (nt == 4) ADVOC_THIN_LAUNCH_PL(4)
  else if (nt == 2) ADVOC_THIN_LAUNCH_PL(2)
  else ADVOC_THIN_LAUNCH_PL(1)
#undef ADVOC_THIN_LAUNCH_PL
#undef ADVOC_THIN_LAUNCH
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  if (fw == -3 && p.ocolsum_out) return launch_colsum_reduce(p.ocolsum_table, p.ocolsum_out, p.d[0].c, stream);
  return ADVOC_OK;
}

// ---------------------------------------------------------------------------------------------
// thin_wgrad
// ---------------------------------------------------------------------------------------------
// One wave streams a run of grid points 16 at a time (8 MFMA steps of 2 points).  The wide
// operand Q is fetched with 16-byte loads (full 128-byte channel rows), parked in a wave-private
// LDS tile [16 points][32 NT channels] and read back in MFMA layout; the next tile's loads are in
// flight while the current one is multiplied.  The thin operand P (<= 2 channels) is gathered
// with scalar loads (it is a small, cache-resident tensor).
constexpr int kWgPatchCols = 15 * 2 + 4;               // 16 grid points at stride <= 2 + tap span
constexpr int kWgPatch = kThinPatchRows * kWgPatchCols * 2;

// PL: P-patch elements per lane, ceil(pr * pc * ca / 64)
template <int NT, int PL>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const WgradParams p, int chunk_tiles, int tiles_x,
                                                         int dy_min, int dx_min, int pr, int pc) {
  constexpr int U = 8;                 // MFMA steps per tile
  constexpr int PT = 2 * U;            // grid points per tile (one grid row, 16 consecutive columns)
  constexpr int LDQ = 32 * NT + 4;
  constexpr int QL = (PT * 8 * NT) / 64;   // float4 slots per lane per tile (= 2 NT)
  __shared__ __attribute__((aligned(16))) float s_q[4][PT * LDQ];
  // the thin operand a tile reads (pr tap rows x pc columns x ca channels), staged per wave with
  // coalesced loads; the per-step operand gathers are ds_reads at tile-invariant offsets
  __shared__ float s_patch[4][kWgPatch];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // uniform for the compiler too (tile ranges, origins: scalar arithmetic)
  const int half = lane >> 5, l32 = lane & 31;
  const int ca = p.P.c0 + p.P.c1;            // 1 or 2
  const int cb = p.Q.c0 + p.Q.c1;
  const int rows = p.ntaps * ca;             // <= 32 live rows of the A operand
  const int b0 = blockIdx.y * (32 * NT);
  const float pslope = p.P.act == ADVOC_ACT_LRELU02 ? 0.2f : (p.P.act == ADVOC_ACT_RELU ? 0.f : 1.f);
  const float qslope = p.Q.act == ADVOC_ACT_LRELU02 ? 0.2f : (p.Q.act == ADVOC_ACT_RELU ? 0.f : 1.f);
  float* Qs = &s_q[wave][0];
  float* patch = &s_patch[wave][0];

  // this lane's A row (tap, a): patch offset of grid point 0 of the tile; step u adds (2u + half) sx ca
  const bool row_ok = l32 < rows;
  int a_off = -1;
  if (row_ok) {
    const int t = l32 / ca, a = l32 % ca;
    const int tp = p.tap[t];
    const int dy = (int)(int8_t)(tp & 0xff), dx = (int)(int8_t)((tp >> 8) & 0xff);
    a_off = ((dy - dy_min) * pc + (dx - dx_min) + half * p.sx) * ca + a;
  }
  const int a_step = 2 * p.sx * ca;

  // patch elements this lane stages: e = lane + 64 i -> (row, col, channel)
  const int patch_elems = pr * pc * ca;
  int pe_row[PL], pe_col[PL], pe_ci[PL];
#pragma unroll
  for (int i = 0; i < PL; ++i) {
    const int e = lane + 64 * i;
    const int ee = e < patch_elems ? e : 0;
    pe_ci[i] = ee % ca;
    const int px = ee / ca;
    pe_col[i] = px % pc;
    pe_row[i] = e < patch_elems ? px / pc : -1;
  }
  // Q loader slots: slot i covers point qk of the tile and channel quad qch
  int qk[QL], qch[QL];
  bool q_on[QL];
#pragma unroll
  for (int i = 0; i < QL; ++i) {
    const int idx = lane + 64 * i;
    qk[i] = idx / (8 * NT);
    qch[i] = b0 + 4 * (idx % (8 * NT));
    q_on[i] = qch[i] < cb;
  }

  const int total_tiles = p.batch * p.gh * tiles_x;
  const int t_begin = (blockIdx.x * 4 + wave) * chunk_tiles;
  const int t_end = t_begin + chunk_tiles < total_tiles ? t_begin + chunk_tiles : total_tiles;

  float4 rq[QL];
  int rq_off[QL];
  unsigned rq_live = 0;
  float pv[PL], pm[PL];
  // Raw loads of tile `TILE` (both operands) into registers; the affine / activation / mask are
  // applied when the tile is parked in LDS, so these stay in flight across the previous MFMAs.
#define ADVOC_TW_FETCH(TILE)                                                                          \
  {                                                                                                   \
    const int rowid_ = (TILE) / tiles_x;                                                              \
    const int gx0_ = ((TILE) - rowid_ * tiles_x) * PT;                                                \
    const int img_ = rowid_ / p.gh, gy_ = rowid_ - img_ * p.gh;                                       \
    rq_live = 0;                                                                                      \
    _Pragma("unroll") for (int i = 0; i < QL; ++i) {                                                  \
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                     \
      rq_off[i] = 0;                                                                                  \
      const int gx = gx0_ + qk[i];                                                                    \
      if (q_on[i] && gx < p.gw) {                                                                     \
        const bool second = qch[i] >= p.Q.c0;                                                         \
        const float* src = second ? p.Q.p1 : p.Q.p0;                                                  \
        const int row = img_ * p.Q.h + gy_;                                                           \
        const int off = second ? (row * p.Q.pitch1 + gx) * p.Q.c1 + (qch[i] - p.Q.c0)                 \
                               : (row * p.Q.pitch0 + gx) * p.Q.c0 + qch[i];                           \
        v = *reinterpret_cast<const float4*>(src + off);                                              \
        rq_off[i] = off;                                                                              \
        rq_live |= 1u << i;                                                                           \
      }                                                                                               \
      rq[i] = v;                                                                                      \
    }                                                                                                 \
    const int iy0_ = gy_ * p.sy + dy_min, ix0_ = gx0_ * p.sx + dx_min;                                \
    _Pragma("unroll") for (int i = 0; i < PL; ++i) {                                                  \
      const int iy = iy0_ + pe_row[i], ix = ix0_ + pe_col[i];                                         \
      float v = 0.f, mk = 0.f;     /* mk doubles as the in-bounds flag: 0 outside the image */        \
      if (pe_row[i] >= 0 && (unsigned)iy < (unsigned)p.P.h && (unsigned)ix < (unsigned)p.P.w) {       \
        const bool second = pe_ci[i] >= p.P.c0;                                                       \
        const int off = second ? ((img_ * p.P.h + iy) * p.P.pitch1 + ix) * p.P.c1 + (pe_ci[i] - p.P.c0) \
                               : ((img_ * p.P.h + iy) * p.P.pitch0 + ix) * p.P.c0 + pe_ci[i];         \
        v = (second ? p.P.p1 : p.P.p0)[off];                                                          \
        mk = (p.P.mask && !second) ? p.P.mask[off] * p.P.mask_scale : 1.f;                            \
      }                                                                                               \
      pv[i] = v;                                                                                      \
      pm[i] = mk;                                                                                     \
    }                                                                                                 \
  }

#define ADVOC_TW_PARK()                                                                               \
  _Pragma("unroll") for (int i = 0; i < QL; ++i) {                                                    \
    float4 v = rq[i];                                                                                 \
    const bool live_ = (rq_live >> i) & 1u;                                                           \
    if (p.Q.scale) {                                                                                  \
      const float4 sc = *reinterpret_cast<const float4*>(p.Q.scale + qch[i]);                         \
      const float4 sh = *reinterpret_cast<const float4*>(p.Q.shift + qch[i]);                         \
      const float k_ = live_ ? 1.f : 0.f;                                                             \
      v.x = v.x * sc.x + sh.x * k_; v.y = v.y * sc.y + sh.y * k_;                                     \
      v.z = v.z * sc.z + sh.z * k_; v.w = v.w * sc.w + sh.w * k_;                                     \
    }                                                                                                 \
    v.x = fmaxf(v.x, qslope * v.x); v.y = fmaxf(v.y, qslope * v.y);                                   \
    v.z = fmaxf(v.z, qslope * v.z); v.w = fmaxf(v.w, qslope * v.w);                                   \
    if (p.Q.mask && live_ && qch[i] < p.Q.c0) {                                                       \
      const uchar4 mk = *reinterpret_cast<const uchar4*>(p.Q.mask + rq_off[i]);                       \
      v.x *= mk.x * p.Q.mask_scale; v.y *= mk.y * p.Q.mask_scale;                                     \
      v.z *= mk.z * p.Q.mask_scale; v.w *= mk.w * p.Q.mask_scale;                                     \
    }                                                                                                 \
    if (q_on[i]) *reinterpret_cast<float4*>(Qs + qk[i] * LDQ + (qch[i] - b0)) = v;                    \
    if (p.qsum_table && live_) { qs[i].x += v.x; qs[i].y += v.y; qs[i].z += v.z; qs[i].w += v.w; }      \
  }                                                                                                   \
  _Pragma("unroll") for (int i = 0; i < PL; ++i) {                                                    \
    if (pe_row[i] < 0) continue;                                                                      \
    float v = pv[i];                                                                                  \
    if (p.P.scale) v = pm[i] != 0.f ? v * p.P.scale[pe_ci[i]] + p.P.shift[pe_ci[i]] : 0.f;            \
    v = fmaxf(v, pslope * v);                                                                         \
    patch[lane + 64 * i] = v * pm[i];                                                                 \
  }

  floatx16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // per-channel sums of the parked Q values (the bias gradient when Q is the output gradient): every Q element passes
  // through exactly one PARK of the launch
  float4 qs[QL];
#pragma unroll
  for (int i = 0; i < QL; ++i) qs[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  if (t_begin < t_end) ADVOC_TW_FETCH(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    ADVOC_TW_PARK();
    wave_lds_sync();
    if (tile + 1 < t_end) ADVOC_TW_FETCH(tile + 1);     // next tile in flight during this tile's MFMAs
    float av[U];
#pragma unroll
    for (int u = 0; u < U; ++u) av[u] = row_ok ? patch[a_off + u * a_step] : 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], Qs[(2 * u + half) * LDQ + 32 * j + l32], acc[j], 0, 0, 0);
    wave_lds_sync();
  }
#undef ADVOC_TW_FETCH
#undef ADVOC_TW_PARK

  // Combine the four waves of the block in LDS, then ONE atomic per output element per block:
  // thousands of waves hammering the same <= 32 x cb addresses serialise in L2 otherwise.
  // (The sums meet in the Q staging area, which is dead by now, two waves at a time: a [4][NT][16][64] array of its own
  // was 64 KB of the 104 KB of the NT = 4 instance -- ONE workgroup per CU, i.e. one wave per SIMD with nothing to run
  // while it waits for its loads.)
  static_assert(2 * NT * 1024 <= 4 * PT * LDQ, "the reduction fits in the staging area");
  __syncthreads();
  float* red = &s_q[0][0];                // [2][NT][16][64]
  if (wave >= 2) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wave - 2) * NT + j) * 1024 + r * 64 + lane] = acc[j][r];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(wave * NT + j) * 1024 + r * 64 + lane] += acc[j][r];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < NT * 16 * 64; idx += 256) {
    const int ln = idx & 63, r = (idx >> 6) & 15, j = idx >> 10;
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);     // A row index (tap, a)
    const int b = b0 + 32 * j + (ln & 31);
    if (row >= rows || b >= cb) continue;
    const float v = red[j * 1024 + r * 64 + ln] + red[(NT + j) * 1024 + r * 64 + ln];
    const int rt = row / ca, ra = row % ca;
    unsafeAtomicAdd(p.dw + ((int64_t)(p.tap[rt] >> 16) * ca + ra) * cb + b, v);
  }
  if (p.qsum_table) {
    // Every loader slot of a lane covers the SAME channel quad (slot i is element lane + 64 i of the tile and 64 is a
    // multiple of the 8 NT quads of a row), and the lanes that share a quad are 8 NT apart: the slots are summed in the
    // lane, the lanes by xor shuffles, the four waves through LDS with plain stores, then ONE global atomic per channel
    // and workgroup into the replica this workgroup belongs to (same-address atomics from every workgroup would queue up in
    // the L2).  (r3 summed the lanes with ds_add_f32, 8 lanes per address: the bursts jammed the CU's LDS pipe badly enough
    // to expose a missing wait in the LDS-DMA kernels running beside this one -- lds_dma.h, dma_ring_barrier -- and cost
    // more than the shuffles.)
    static_assert(64 % (8 * NT) == 0, "a lane's loader slots share their channel quad");
    float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < QL; ++i)
      if (q_on[i]) { tot.x += qs[i].x; tot.y += qs[i].y; tot.z += qs[i].z; tot.w += qs[i].w; }
#pragma unroll
    for (int off = 8 * NT; off < 64; off <<= 1) {
      tot.x += __shfl_xor(tot.x, off, 64); tot.y += __shfl_xor(tot.y, off, 64);
      tot.z += __shfl_xor(tot.z, off, 64); tot.w += __shfl_xor(tot.w, off, 64);
    }
    __syncthreads();
    float* s_sum = &s_patch[0][0];               // [4 waves][32 NT]
    static_assert(4 * 32 * NT <= 4 * kWgPatch, "the per-wave sums fit in the patch staging area");
    if (lane < 8 * NT) *reinterpret_cast<float4*>(s_sum + wave * (32 * NT) + 4 * lane) = tot;
    __syncthreads();
    for (int t = threadIdx.x; t < 32 * NT; t += 256)
      if (b0 + t < cb)
        unsafeAtomicAdd(p.qsum_table + (size_t)(blockIdx.x & (kColsumReplicas - 1)) * cb + b0 + t,
                        (s_sum[t] + s_sum[32 * NT + t]) + (s_sum[2 * 32 * NT + t] + s_sum[3 * 32 * NT + t]));
  }
}

}  // namespace

int launch_thin_k_gemm(const GatherGemmParams& p, bool b_kn, hipStream_t stream, const char** name_only) {
  const int ktot = p.c0 + p.c1;
  if (ktot < 1 || ktot > 2 || p.c0 > 2 || p.n_total % 32) return ADVOC_ERR_UNSUPPORTED;
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
  int nt = cb % 128 == 0 ? 4 : (cb % 64 == 0 ? 2 : 1);
  if (nt > tuning().thin_wgrad_nt) nt = tuning().thin_wgrad_nt;
  if (name_only) {
    *name_only = nt == 4 ? "thin_wgrad_kernel<4>" : (nt == 2 ? "thin_wgrad_kernel<2>" : "thin_wgrad_kernel<1>");
    return ADVOC_OK;
  }
  if (!p.accumulate) {
    hipError_t e = hipMemsetAsync(p.dw, 0, sizeof(float) * (size_t)p.ntaps * ca * cb, stream);
    if (e != hipSuccess) { note_hip_error(e); return ADVOC_ERR_HIP; }
  }
  if (p.qsum_table) {
    hipError_t e = hipMemsetAsync(p.qsum_table, 0, sizeof(float) * kColsumReplicas * (size_t)cb, stream);
    if (e != hipSuccess) { note_hip_error(e); return ADVOC_ERR_HIP; }
  }
  int dy_min = 127, dy_max = -128, dx_min = 127, dx_max = -128;
  for (int t = 0; t < p.ntaps; ++t) {
    const int dy = (int)(int8_t)(p.tap[t] & 0xff), dx = (int)(int8_t)((p.tap[t] >> 8) & 0xff);
    dy_min = dy < dy_min ? dy : dy_min; dy_max = dy > dy_max ? dy : dy_max;
    dx_min = dx < dx_min ? dx : dx_min; dx_max = dx > dx_max ? dx : dx_max;
  }
  const int pr = dy_max - dy_min + 1, pc = 15 * p.sx + (dx_max - dx_min) + 1;
  if (pr > kThinPatchRows || pc > kWgPatchCols) return ADVOC_ERR_UNSUPPORTED;
  const int pl = (pr * pc * ca + 63) / 64;
  const int tiles_x = (p.gw + 15) / 16;
  const int64_t tiles = (int64_t)p.batch * p.gh * tiles_x;
  if (tiles > 0x7fffffffLL / 64) return ADVOC_ERR_UNSUPPORTED;
  const int by = cb / (32 * nt);
  // one round of the chip (as many workgroups as are resident at once: occupancy query per instance, cached), the tile
  // axis cut into one contiguous run per wave; each wave keeps one tile of loads in flight while it multiplies the
  // previous one
  dim3 grid(1, (unsigned)by);
  int64_t chunk = 4;
  ADVOC_CLEAR_LAUNCH_ERROR();
#define ADVOC_TW_LAUNCH(NT_, PL_)                                                                     \
  {                                                                                                   \
    static int per_cu = 0;                                                                            \
    if (per_cu == 0) {                                                                                \
      int nb = 0;                                                                                     \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, thin_wgrad_kernel<NT_, PL_>, 256, 0) != hipSuccess || nb < 1) \
        nb = 1;                                                                                       \
      per_cu = nb;                                                                                    \
    }                                                                                                 \
    int64_t waves = (int64_t)4 * per_cu * device_cu_count() / by;                                     \
    if (waves < 4) waves = 4;                                                                         \
    chunk = ceil_div(tiles, waves);                                                                   \
    if (chunk < 4) chunk = 4;                                                                         \
    grid.x = (unsigned)ceil_div(ceil_div(tiles, chunk), 4);                                           \
    hipLaunchKernelGGL((thin_wgrad_kernel<NT_, PL_>), grid, dim3(256), 0, stream, p, (int)chunk, tiles_x, dy_min, dx_min, \
                       pr, pc);                                                                       \
  }
#define ADVOC_TW_LAUNCH_PL(NT_)                                                                       \
  {                                                                                                   \
    if (pl <= 2) ADVOC_TW_LAUNCH(NT_, 2)                                                              \
    else if (pl <= 3) ADVOC_TW_LAUNCH(NT_, 3)                                                         \
    else ADVOC_TW_LAUNCH(NT_, 5)                                                                      \
  }
  if (nt == 4) ADVOC_TW_LAUNCH_PL(4)
  else if (nt == 2) ADVOC_TW_LAUNCH_PL(2)
  else ADVOC_TW_LAUNCH_PL(1)
#undef ADVOC_TW_LAUNCH_PL
#undef ADVOC_TW_LAUNCH
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  if (p.qsum_table && p.qsum_out) return launch_colsum_reduce(p.qsum_table, p.qsum_out, cb, stream);
  return ADVOC_OK;
}

}  // namespace advoc
