// Direct kernels for the thin edge layers of the AdVoc nets (gfx950): 1-2 input channels
// (generator encoder_1, discriminator layer_1) or a single output channel (generator decoder_1,
// discriminator layer_5).  Arithmetic intensity is 7-14 flop/byte: these
// are HBM-bound, so no MFMA -- coalesced channel-contiguous accesses, operands broadcast from
// L1, wave-shuffle reductions.  They consume the same GatherGemmParams as the MFMA kernel.
//
// Reference ops replaced: Conv2D / Conv2DBackpropInput for advoc_model.py:91-94 (encoder_1),
// :153-158 (decoder_1), :185-188 (layer_1), :199-202 (layer_5) and their gradients.
#include <stdlib.h>

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


// ---------------------------------------------------------------------------------------------
// fused_taps (r4): the two-stage path in ONE launch.  <= 2 output columns over a wide K used to run as a pointwise GEMM
// S[input pixel][tap * N + n] = sum_k act(A[pixel, k]) w[tap][n][k] written to the workspace, then tap_sum_kernel: S (16 or
// 32 floats per input pixel) went to HBM and came back (1.37 x the algorithmic traffic, 1.4 ms per AdVoc-full step in two
// launches per layer).  Here a workgroup owns a TY x TX patch of grid points of one image: it streams the patch's input
// region (halo included) ONCE -- 16 pixels per wave at a time, every lane a 16-byte piece of its pixel's channels, the
// contraction on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: pixels x taps) with the contraction order permuted to the
// load layout (lane (pixel, g) holds k = 16 s + 4 g + 0..3; the weight fragments sit in LDS in the same order) -- keeps S
// for the region in LDS, and sums the taps from there with tap_sum_kernel's epilogue.  S never exists in HBM; the halo
// (one or two input rows / columns per patch side) is read again by the neighbouring patch, from L2 / the memory-side
// cache as a rule.  Loads of the next 16-pixel group are in flight under the current group's MFMAs.
// ---------------------------------------------------------------------------------------------
typedef float f4v __attribute__((ext_vector_type(4)));

struct FusedGeom {
  int dy_min, dx_min;
  int ry, rx;           // input region of a patch (rows, columns), rx_magic = ceil(2^20 / rx)
  int rx_magic;
  int ty, tx;           // grid points per patch
  int tiles_y, tiles_x;
  int ld;               // words per region pixel in LDS (columns + 1: conflict-free gathers)
  int w_floats;         // floats of weight fragments at the start of the dynamic LDS
  int n_eff, n0;        // columns per tap the launch computes and the first of them: a two-column problem one of whose
                        // destinations is null (the discriminator's conditioning channel in the G step) runs as one column
  int pairs;            // != 0: the four phases are the 2 x 2 sub-pixels of a stride-2 result on one-channel destinations without
                        // gradient gating: phase B stores the two pixels of a row together
};

template <int CB, int KS>     // CB column blocks of 16 (taps x N = 16 CB); KS k steps of 16 channels per load group
__global__ __launch_bounds__(256, 2) void fused_taps_kernel(const GatherGemmParams p, const FusedGeom g) {
  extern __shared__ __attribute__((aligned(16))) float fsm[];
  __shared__ int s_tap[kMaxPhases][kMaxTaps];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // uniform for the compiler too: the loop counters derived from it
                                                                   // select between the two source descriptors
  const int K = p.c0 + p.c1, nsteps = K >> 4;
  float* const Wl = fsm;                           // [CB][K / 16][64 lanes][4]
  float* const aff = fsm + g.w_floats;             // [2][K] scale, shift (when the input has an affine)
  float* const S = aff + (p.in_scale ? 2 * K : 0); // [ry * rx][ld]
  if (tid < kMaxPhases * kMaxTaps) s_tap[tid / kMaxTaps][tid % kMaxTaps] = p.tap[tid / kMaxTaps][tid % kMaxTaps];
  for (int i = tid; i < CB * nsteps * 64; i += 256) {
    const int l = i & 63, st = (i >> 6) % nsteps, cb = (i >> 6) / nsteps;
    const int col = 16 * cb + (l & 15), k = 16 * st + 4 * (l >> 4);
    const int wcol = (col / g.n_eff) * p.n_total + g.n0 + col % g.n_eff;          // [tap][n] of the weight tensor
    *reinterpret_cast<float4*>(Wl + 4 * i) = *reinterpret_cast<const float4*>(p.w + (int64_t)wcol * K + k);
  }
  if (p.in_scale)
    for (int i = tid; i < K; i += 256) { aff[i] = p.in_scale[i]; aff[K + i] = p.in_shift[i]; }

  int b = blockIdx.x;
  const int tx_i = b % g.tiles_x; b /= g.tiles_x;
  const int ty_i = b % g.tiles_y;
  const int img = b / g.tiles_y;
  const int gy0 = ty_i * g.ty, gx0 = tx_i * g.tx;
  const int iy0 = gy0 * p.sy + g.dy_min, ix0 = gx0 * p.sx + g.dx_min;
  const int R = g.ry * g.rx;
  const int groups = (R + 15) >> 4;
  const int nk = nsteps / KS;                      // load groups per 16 pixels
  const int kg = lane >> 4, prow = lane & 15;
  const float slope = p.in_act == ADVOC_ACT_LRELU02 ? 0.2f : (p.in_act == ADVOC_ACT_RELU ? 0.f : 1.f);
  __syncthreads();

  // ---- phase A: S = act(A) W for the region, 16 pixels per wave and step ----
  // Loads are buffer instructions: a lane whose pixel lies outside the region or the image gets an offset beyond the
  // descriptor's range and reads zeros -- no branch around any load (with `in ? load : 0` every load sat in its own
  // exec-masked block and the waits at the joins were vmcnt(0): nothing stayed in flight under the MFMAs).
  const int my_groups = groups > wave ? (groups - wave + 3) >> 2 : 0;
  const int n_it = my_groups * nk;
  constexpr unsigned kOob = 0xffffff00u;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a0), 0, kOob, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.c1 ? p.a1 : p.a0), 0, p.c1 ? kOob : 0u, 0x00020000);
  const int steps0 = p.c0 >> 4;                   // k steps that read source 0
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 bufA[KS], bufB[KS];
  float liveA = 0.f, liveB = 0.f;      // 1 where the lane's pixel is inside the image: the zero padding applies to the TRANSFORMED input
  f4v acc[CB];
  int l_gi = wave, l_kc = 0, c_gi = wave, c_kc = 0;        // (16-pixel group, load group) of the next load / compute call
  auto load = [&](u32x4 (&buf)[KS], float& live) {
    const int q = 16 * l_gi + prow;
    const int py = (q * g.rx_magic) >> 20, px = q - py * g.rx;
    const int iy = iy0 + py, ix = ix0 + px;
    const bool in = l_gi < groups && q < R && (unsigned)iy < (unsigned)p.in_h && (unsigned)ix < (unsigned)p.in_w;
    const int rowi = img * p.a_h + iy;
    live = in ? 1.f : 0.f;
    const unsigned o0 = in ? (unsigned)((rowi * p.a0_pitch + ix) * p.c0 + 4 * kg) * 4u : kOob;
    const unsigned o1 = in ? (unsigned)((rowi * p.a1_pitch + ix) * p.c1 + 4 * kg) * 4u : kOob;
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      const int st = l_kc * KS + u;
      const bool first = st < steps0;               // wave-uniform: c0 is a multiple of 16
      const unsigned off = first ? o0 + 64u * st : o1 + 64u * (st - steps0);
      buf[u] = __builtin_amdgcn_raw_buffer_load_b128(first ? rs0 : rs1, (in ? off : kOob), 0, 0);
    }
    if (++l_kc == nk) { l_kc = 0; l_gi += 4; }
  };
  auto compute = [&](const u32x4 (&buf)[KS], float live) {
    if (c_kc == 0) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) acc[cb] = (f4v){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      const int st = c_kc * KS + u;
      f4v a = {__uint_as_float(buf[u].x), __uint_as_float(buf[u].y), __uint_as_float(buf[u].z), __uint_as_float(buf[u].w)};
      if (p.in_scale) {
        const f4v sc = *reinterpret_cast<const f4v*>(aff + 16 * st + 4 * kg);
        const f4v sh = *reinterpret_cast<const f4v*>(aff + K + 16 * st + 4 * kg);
        a = (a * sc + sh) * live;          // pixels outside the image are zeros AFTER the affine
      }
      a.x = fmaxf(a.x, slope * a.x); a.y = fmaxf(a.y, slope * a.y);
      a.z = fmaxf(a.z, slope * a.z); a.w = fmaxf(a.w, slope * a.w);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const f4v w = *reinterpret_cast<const f4v*>(Wl + 4 * ((cb * nsteps + st) * 64 + lane));
        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.x, acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.y, acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.z, acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.w, acc[cb], 0, 0, 0);
      }
    }
    if (c_kc == nk - 1) {    // accumulator element i of a lane: pixel 4 (lane / 16) + i of the group, column lane % 16
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int q = 16 * c_gi + 4 * kg + i;
          if (q < R) S[q * g.ld + 16 * cb + prow] = acc[cb][i];
        }
    }
    if (++c_kc == nk) { c_kc = 0; c_gi += 4; }
  };
  // every load call is UNCONDITIONAL (beyond the wave's last group all lanes are out of range: no memory traffic): with
  // `if (more) load(...)` the compiler has to assume the smaller number of loads in flight at each wait, i.e. it waits for
  // the loads it has just issued as well
  if (n_it > 0) {
    load(bufA, liveA);
    for (int it = 0; it < n_it; it += 2) {
      load(bufB, liveB);
      compute(bufA, liveA);
      load(bufA, liveA);
      if (it + 1 < n_it) compute(bufB, liveB);
    }
  }
  __syncthreads();

  // ---- phase B: out[g, n] = bias[n] + sum over the taps of S[g * s + d_t][wtap_t * N + n] (tap_sum_kernel's) ----
  const int N = g.n_eff;
  if (g.pairs) {
    // four sub-pixel phases (0,0) (0,1) (1,0) (1,1) of a stride-2 result: a grid point owns a 2 x 2 block of output pixels per
    // column -- the two pixels of a row go out as ONE 8-byte store (the destinations of these layers have one channel),
    // bias and accumulation applied to both; 16 taps x N values are summed before anything is stored
    // (four taps per phase: the 16 LDS offsets are wave-uniform and computed once, from the kernel argument)
    int toff[4][4];
#pragma unroll
    for (int phase = 0; phase < 4; ++phase)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int tp = p.tap[phase][t];
        toff[phase][t] = (((int)(int8_t)(tp & 0xff) - g.dy_min) * g.rx + (int)(int8_t)((tp >> 8) & 0xff) - g.dx_min) * g.ld + (tp >> 16) * N;
      }
    for (int pt = tid; pt < g.ty * g.tx; pt += 256) {
      const int gyl = pt / g.tx, gxl = pt - gyl * g.tx;
      const int gy = gy0 + gyl, gx = gx0 + gxl;
      if (gy >= p.gh || gx >= p.gw) continue;
      const float* const base = S + (gyl * p.sy * g.rx + gxl * p.sx) * g.ld;
      float v[4][2];
#pragma unroll
      for (int phase = 0; phase < 4; ++phase) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float* row = base + toff[phase][t];
          a0 += row[0];
          if (N > 1) a1 += row[1];
        }
        v[phase][0] = a0; v[phase][1] = a1;
      }
      const int ox = 2 * gx;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        if (n >= N) break;
        const GemmDest& d = p.d[g.n0 + n >= p.n_split ? 1 : 0];
        if (d.p == nullptr) continue;
        const float bias = p.bias ? p.bias[g.n0 + n] : 0.f;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
          const int oy = 2 * gy + py;
          if (oy >= p.out_h || ox >= p.out_w) continue;
          float* dst = d.p + ((int64_t)img * p.out_h + oy) * d.pitch + ox;       // one channel per destination
          float2 r = make_float2(v[2 * py][n] + bias, v[2 * py + 1][n] + bias);
          if (ox + 1 < p.out_w) {         // 4-byte aligned only (odd pitches): global dwordx2 accesses take that
            // (a plain two-float vector with 4-byte alignment: HIP's float2 class takes 8-byte aligned `this`, and
            // -Walign-mismatch says so at every instantiation)
            typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
            f32x2_a4* const d2 = reinterpret_cast<f32x2_a4*>(dst);
            if (d.accum) { const f32x2_a4 o = *d2; r.x += o.x; r.y += o.y; }
            f32x2_a4 rv;
            rv.x = r.x; rv.y = r.y;
            *d2 = rv;
          } else {
            dst[0] = d.accum ? r.x + dst[0] : r.x;
          }
        }
      }
    }
    return;
  }
  for (int pt = tid; pt < g.ty * g.tx; pt += 256) {
    const int gyl = pt / g.tx, gxl = pt - gyl * g.tx;
    GridPoint gp;
    gp.img = img;
    gp.gy = gy0 + gyl;
    gp.gx = gx0 + gxl;
    if (gp.gy >= p.gh || gp.gx >= p.gw) continue;
    const int ly = gyl * p.sy - g.dy_min, lx = gxl * p.sx - g.dx_min;
    for (int phase = 0; phase < p.nphase; ++phase) {
      float acc0 = 0.f, acc1 = 0.f;
      for (int t = 0; t < p.ntaps; ++t) {
        const int tp = s_tap[phase][t];
        const int py = ly + (int)(int8_t)(tp & 0xff);
        const int px = lx + (int)(int8_t)((tp >> 8) & 0xff);
        const float* row = S + (py * g.rx + px) * g.ld + (tp >> 16) * N;
        acc0 += row[0];
        if (N > 1) acc1 += row[1];
      }
      store_result(p, phase, gp, g.n0, acc0);
      if (N > 1) store_result(p, phase, gp, g.n0 + 1, acc1);
    }
  }
}

}  // namespace

// The fused form of the two-stage path (fused_taps_kernel): shapes it takes, the patch geometry it picks.
static bool fused_taps_plan(const GatherGemmParams& p, FusedGeom* out, int* cb_out, int* ks_out, size_t* lds_out) {
  const int K = p.c0 + p.c1;
  if (p.n_total < 1 || p.n_total > 2) return false;
  // columns whose destination exists (backward-data of the discriminator's first layer in the G step: the conditioning
  // channel's gradient is not wanted)
  int n0 = 0, N = p.n_total;
  if (N == 2 && p.n_split == 1) {
    if (p.d[0].p == nullptr && p.d[1].p != nullptr) { n0 = 1; N = 1; }
    else if (p.d[1].p == nullptr && p.d[0].p != nullptr) { N = 1; }
  }
  const int cols = p.nphase * p.ntaps * N;
  if ((cols != 16 && cols != 32) || K % 32 || p.c0 % 16 || K > 1024 || p.a_mask || p.y_mask) return false;
  if ((int64_t)p.batch * p.a_h * p.a0_pitch * p.c0 * 4 >= 0xffffff00LL || (int64_t)p.batch * p.a_h * p.a1_pitch * p.c1 * 4 >= 0xffffff00LL)
    return false;                                    // 32-bit byte offsets into the sources
  int dy_min = 127, dy_max = -128, dx_min = 127, dx_max = -128;
  for (int ph = 0; ph < p.nphase; ++ph)
    for (int t = 0; t < p.ntaps; ++t) {
      const int dy = (int)(int8_t)(p.tap[ph][t] & 0xff), dx = (int)(int8_t)((p.tap[ph][t] >> 8) & 0xff);
      if (dy == -128) return false;                  // padded phases (odd kernels): direct kernel
      dy_min = dy < dy_min ? dy : dy_min; dy_max = dy > dy_max ? dy : dy_max;
      dx_min = dx < dx_min ? dx : dx_min; dx_max = dx > dx_max ? dx : dx_max;
    }
  const int ld = cols + 1;
  const int w_floats = (cols / 16) * (K / 16) * 256;
  const size_t fixed = sizeof(float) * ((size_t)w_floats + (p.in_scale ? 2 * (size_t)K : 0));
  constexpr int budget_kb = 78;      // two workgroups per CU (r4 A/B: ADVOC_FUSED_TAPS_LDS_KB, removed in r6)
  const size_t budget = (size_t)(budget_kb < 16 ? 16 : (budget_kb > 156 ? 156 : budget_kb)) * 1024;
  if (fixed + 64 * sizeof(float) * ld > budget) return false;
  const int64_t rmax = (int64_t)((budget - fixed) / (sizeof(float) * ld));
  // patch = TY x TX grid points, region = ((TY - 1) sy + span_y + 1) x ((TX - 1) sx + span_x + 1) input pixels <= rmax:
  // the shape that streams the fewest input pixels over the whole grid (partial patches at the edges included)
  const int span_y = dy_max - dy_min, span_x = dx_max - dx_min;
  int64_t best = -1;
  FusedGeom g = {};
  for (int ty = 1; ty <= p.gh && ty <= 128; ++ty) {
    const int ry = (ty - 1) * p.sy + span_y + 1;
    if (ry > rmax) break;
    const int64_t rx_max = rmax / ry;
    int tx = (int)((rx_max - span_x - 1) / p.sx) + 1;
    if (rx_max < span_x + 1 || tx < 1) break;
    if (tx > p.gw) tx = p.gw;
    const int rx = (tx - 1) * p.sx + span_x + 1;
    if (rx > 4096) continue;
    const int tiles_y = (int)ceil_div(p.gh, ty), tiles_x = (int)ceil_div(p.gw, tx);
    const int64_t cost = (int64_t)tiles_y * tiles_x * ry * rx;
    if (best < 0 || cost < best) {
      best = cost;
      g.ry = ry; g.rx = rx; g.ty = ty; g.tx = tx; g.tiles_y = tiles_y; g.tiles_x = tiles_x;
    }
  }
  if (best < 0) return false;
  g.dy_min = dy_min; g.dx_min = dx_min; g.ld = ld; g.w_floats = w_floats;
  g.n_eff = N; g.n0 = n0;
  g.rx_magic = ((1 << 20) + g.rx - 1) / g.rx;
  // q / rx == (q * rx_magic) >> 20 for every q the kernel divides (q < R rounded up to its 16-row slab) iff
  // q * (rx_magic * rx - 2^20) < 2^20; and the int product must not wrap.  Otherwise: the direct kernel.
  {
    const int64_t qmax = (int64_t)g.ry * g.rx + 15;
    if (qmax * ((int64_t)g.rx_magic * g.rx - (1 << 20)) >= (1 << 20) || qmax * g.rx_magic >= ((int64_t)1 << 31)) return false;
  }
  {
    bool pr = p.nphase == 4 && p.ntaps == 4 && p.osy == 2 && p.osx == 2 && p.grad_act == ADVOC_ACT_NONE;
    for (int ph = 0; pr && ph < 4; ++ph) pr = p.ooy[ph] == (ph >> 1) && p.oox[ph] == (ph & 1);
    for (int n = 0; pr && n < N; ++n) {
      const GemmDest& d = p.d[n0 + n >= p.n_split ? 1 : 0];
      pr = (d.p == nullptr || (d.c == 1 && d.gmask == nullptr)) && (p.n_total == 1 || p.n_split == 1);
    }
    g.pairs = pr ? 1 : 0;
  }
  if ((int64_t)p.batch * g.tiles_y * g.tiles_x > 0x7fffffffLL) return false;
  *out = g;
  *cb_out = cols / 16;
  *ks_out = K % 128 == 0 ? 8 : (K % 64 == 0 ? 4 : 2);
  *lds_out = fixed + sizeof(float) * (size_t)g.ry * g.rx * ld;
  return true;
}

bool fused_taps_ok(const GatherGemmParams& p) {
  FusedGeom g; int cb, ks; size_t lds;
  return fused_taps_plan(p, &g, &cb, &ks, &lds);
}

int launch_fused_taps(const GatherGemmParams& p, hipStream_t stream, const char** name_only) {
  FusedGeom g; int cb, ks; size_t lds;
  if (!fused_taps_plan(p, &g, &cb, &ks, &lds)) return ADVOC_ERR_UNSUPPORTED;
  if (name_only) {
    *name_only = cb == 1 ? (ks == 8 ? "fused_taps_kernel<1, 8>" : ks == 4 ? "fused_taps_kernel<1, 4>" : "fused_taps_kernel<1, 2>")
                         : (ks == 8 ? "fused_taps_kernel<2, 8>" : ks == 4 ? "fused_taps_kernel<2, 4>" : "fused_taps_kernel<2, 2>");
    return ADVOC_OK;
  }
  const void* fn = cb == 1 ? (ks == 8 ? (const void*)fused_taps_kernel<1, 8> : ks == 4 ? (const void*)fused_taps_kernel<1, 4> : (const void*)fused_taps_kernel<1, 2>)
                           : (ks == 8 ? (const void*)fused_taps_kernel<2, 8> : ks == 4 ? (const void*)fused_taps_kernel<2, 4> : (const void*)fused_taps_kernel<2, 2>);
  if (lds > 64 * 1024) {
    const hipError_t attr = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    if (attr != hipSuccess) { note_hip_error(attr); return ADVOC_ERR_HIP; }
  }
  const dim3 grid((unsigned)((int64_t)p.batch * g.tiles_y * g.tiles_x));
  ADVOC_CLEAR_LAUNCH_ERROR();
  if (cb == 1 && ks == 8) hipLaunchKernelGGL((fused_taps_kernel<1, 8>), grid, dim3(256), lds, stream, p, g);
  else if (cb == 1 && ks == 4) hipLaunchKernelGGL((fused_taps_kernel<1, 4>), grid, dim3(256), lds, stream, p, g);
  else if (cb == 1) hipLaunchKernelGGL((fused_taps_kernel<1, 2>), grid, dim3(256), lds, stream, p, g);
  else if (ks == 8) hipLaunchKernelGGL((fused_taps_kernel<2, 8>), grid, dim3(256), lds, stream, p, g);
  else if (ks == 4) hipLaunchKernelGGL((fused_taps_kernel<2, 4>), grid, dim3(256), lds, stream, p, g);
  else hipLaunchKernelGGL((fused_taps_kernel<2, 2>), grid, dim3(256), lds, stream, p, g);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

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
