// Gather-GEMM on operand images: the split-bf16 convolution kernel whose operands arrive PRE-SPLIT.
//
// Same GEMM view as igemm.hip (rows = output grid points, cols = output channels, depth = (tap, input
// channel); replaces the cuDNN / Eigen Conv2D and Conv2DBackpropInput kernels TF1 runs for
// models/advoc/advoc_model.py:25-69 in both directions) and the same arithmetic as its register-split
// variant (x6.h: six bf16 MFMA products per fp32 product, fp32 accumulation, smallest terms first -- results
// are bit-identical to that kernel for the same K order), but the instruction stream around the matrix
// cores is gone:
//   * A is read from an activation image (image.hip: act / BN affine / dropout already applied, ONCE per
//     element instead of once per tap and column tile), B from the weight image.  Both images are arrays of
//     96-byte K SLICES: the three bf16 planes of 16 consecutive contraction slots side by side -- one row of a
//     K tile is ONE contiguous 96-byte piece;
//   * both go global -> LDS by `buffer_load_dwordx4 ... lds` (LDS-DMA): no VGPR staging, no ds_write, no
//     split arithmetic.  The hardware range check of the buffer descriptor supplies the zero padding: a tap
//     that falls outside the input gets an offset beyond num_records and the DMA writes zeros;
//   * one DMA instruction = 8 tile rows x 6 sixteen-byte chunks on lanes 0..47 (768 dense bytes of LDS); a wave
//     issues BM / 32 of them per K tile for A and BN / 32 for B, each lane keeping ONE (row, chunk) per
//     instruction slot for the whole K loop: per K tile the address work is two compares and a select per slot,
//     everything else is scalar;
//   * LDS rows are 96 bytes ([plane][16 slots]); the two 16-byte halves of every plane are stored swapped in
//     odd 8-row blocks (applied to the SOURCE address, the DMA destination is lane-linear), which makes the MFMA
//     fragment reads (ds_read_b128, one row per lane) conflict-free;
//   * NS LDS stages (2 or 3), ONE s_barrier per K tile: wait for the own DMAs of tile t (counted vmcnt, the
//     younger tiles stay in flight) -> barrier -> issue tile t + NS - 1 into the stage tile t - 1 just
//     vacated -> 12 ds_read_b128 + 24 MFMAs (128 x 128 tile) of tile t.
// Tile mapping (XCD-aware order), split-K, the tail split and the epilogue are those of igemm.hip.
#include <stdlib.h>

#include <string>

#include "common.h"
#include "igemm.h"
#include "tuning.h"
#include "x6.h"

namespace advoc {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_void_p;

__device__ __forceinline__ float act_slope(int act) {
  return act == ADVOC_ACT_LRELU02 ? 0.2f : (act == ADVOC_ACT_RELU ? 0.f : 1.f);
}

template <int MT, int NT, int NS>
struct DCfg {
  static constexpr int WGM = 2, WGN = 2;
  static constexpr int BM = 32 * MT * WGM, BN = 32 * NT * WGN;
  static constexpr int BK = 16;
  static constexpr int ROWB = 6 * BK;                       // bytes of one tile row: 3 planes x 16 bf16
  static constexpr int A_TILE = BM * ROWB, B_TILE = BN * ROWB;
  static constexpr int STAGE = A_TILE + B_TILE;             // bytes
  static constexpr int RGA = BM / 32, CGB = BN / 32;        // 8-row DMA blocks per wave and K tile
  static constexpr int DMA_PER_TILE = RGA + CGB;
  static constexpr int EPI_BYTES = 4 * 32 * 36 * 4 + 2 * BM * 4;
  static constexpr size_t LDS_BYTES = NS * STAGE > EPI_BYTES ? NS * STAGE : EPI_BYTES;
  static_assert(MT % 2 == 0 && NT % 2 == 0, "whole 32-row DMA groups per wave");
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// launch_bounds: 8 accumulators (128 x 256 / 256 x 128) need ~200 registers -> 2 waves per SIMD; the 128 x 128
// tile fits 3 (<= 168 registers) when its LDS does (2 stages).
// (the body is a __device__ function behind a one-line kernel: with the body inside the __global__ template itself
// hipcc's HOST pass silently dropped the kernel handle -- no diagnostic, an undefined symbol at load time)
template <int MT, int NT, int NS, int ABL = 0>
__device__ __forceinline__ void gather_gemm_x6d_body(const GatherGemmParams& p) {
  using C = DCfg<MT, NT, NS>;
  constexpr int BM = C::BM, BN = C::BN, BK = C::BK, WGN = C::WGN;
  constexpr int RGA = C::RGA, CGB = C::CGB;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* smem_b = reinterpret_cast<unsigned char*>(smem);
  int* s_pix = reinterpret_cast<int*>(smem + 4 * 32 * 36);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const unsigned M = (unsigned)p.batch * (unsigned)p.gh * (unsigned)p.gw;
  const int ntn = p.n_total / BN;
  int tile, phase, ks_idx = blockIdx.y, ks_cnt = gridDim.y, tail_tile = -1;
  {
    const bool tail_mode = p.tail_main > 0;
    const int nb = tail_mode ? p.tail_main : (int)gridDim.x, b = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = b & 7, slot = b >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;   // bijective for any nb
    phase = blockIdx.z;
    if (tail_mode) {
      if (b >= nb) {
        const int piece = b - nb;
        tail_tile = piece / p.tail_split;
        ks_idx = piece - tail_tile * p.tail_split;
        ks_cnt = p.tail_split;
        tile = nb + tail_tile;
      }
      const int tpp = (int)((M + BM - 1) / BM) * ntn;   // tiles per phase
      phase = tile / tpp;
      tile -= phase * tpp;
    }
  }
  const unsigned m0 = (unsigned)(tile / ntn) * BM;
  const int n0 = (tile % ntn) * BN;
  const int ktot = p.c0 + p.c1;
  const int kpt = ktot / BK;
  const int nkt = kpt * p.ntaps;

  // ---- DMA lanes 0..47: lane -> (row lane / 6 of an 8-row block, 16-byte chunk lane % 6 of the 96-byte row); odd
  // blocks hold every plane's two halves swapped (chunk ^ 1) ----
  const int lrow = lane / 6, lchunk = lane - 6 * lrow;
  const bool dma_lane = lane < 48;
  int a_y[RGA], a_x[RGA], a_b0[RGA], a_b1[RGA];
  bool a_ok[RGA];
#pragma unroll
  for (int g = 0; g < RGA; ++g) {
    const int blk = wave * RGA + g;
    const int r = blk * 8 + lrow;
    const unsigned m = m0 + r;
    a_ok[g] = dma_lane && m < M;
    const unsigned mm = a_ok[g] ? m : 0u;
    const unsigned t = mm / (unsigned)p.gw;
    const int gx = (int)(mm - t * (unsigned)p.gw);
    const int img = (int)(t / (unsigned)p.gh);
    const int gy = (int)(t - (unsigned)img * (unsigned)p.gh);
    a_x[g] = gx * p.sx;
    a_y[g] = gy * p.sy;
    const int gc = lchunk ^ (blk & 1);
    a_b0[g] = (((img * p.a_h + a_y[g]) * p.a0_pitch + a_x[g]) * p.c0) * 6 + gc * 16;     // bytes into the image
    a_b1[g] = (((img * p.a_h + a_y[g]) * p.a1_pitch + a_x[g]) * p.c1) * 6 + gc * 16;
  }
  int b_off[CGB];
#pragma unroll
  for (int g = 0; g < CGB; ++g) {
    const int blk = wave * CGB + g;
    const int n = blk * 8 + lrow;
    const int gc = lchunk ^ (blk & 1);
    b_off[g] = ((n0 + n) * ktot) * 6 + gc * 16;
  }
  // buffer descriptors over the whole images; the weight-slab / K-slice offsets go into soffset
  const __amdgpu_buffer_rsrc_t rs_a0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(p.a0_img), 0, p.a0_img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(p.a1_img ? p.a1_img : p.a0_img), 0, p.a1_img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(p.wq), 0, p.wq_taps * p.n_total * ktot * 6, 0x00020000);

  const int kt_begin = (int)((int64_t)nkt * ks_idx / ks_cnt);
  const int kt_end = (int)((int64_t)nkt * (ks_idx + 1) / ks_cnt);
  const bool taps_inner = p.k_order != 0;
  int ld_tap = taps_inner ? kt_begin % p.ntaps : kt_begin / kpt;
  int ld_k0 = (taps_inner ? kt_begin / p.ntaps : kt_begin % kpt) * BK;

  // Issues the DMAs of the next K tile of the walk into stage `st` (a compile-time constant in the unrolled
  // loop below).  Tiles past kt_end wrap round to valid ones; their data is never read.
#define ADVOC_X6D_ISSUE(ST)                                                                              \
  {                                                                                                      \
    const int ti_ = ld_tap;                                                                              \
    const int k0_ = ld_k0;                                                                               \
    if (taps_inner) {                                                                                    \
      if (++ld_tap == p.ntaps) { ld_tap = 0; ld_k0 += BK; if (ld_k0 == ktot) ld_k0 = 0; }                \
    } else {                                                                                             \
      ld_k0 += BK;                                                                                       \
      if (ld_k0 == ktot) { ld_k0 = 0; if (++ld_tap == p.ntaps) ld_tap = 0; }                             \
    }                                                                                                    \
    const int tp_ = __builtin_amdgcn_readfirstlane(p.tap[phase][ti_]);                                   \
    const int dy_ = (int)(int8_t)(tp_ & 0xff), dx_ = (int)(int8_t)((tp_ >> 8) & 0xff);                   \
    const int wtap_ = tp_ >> 16;                                                                         \
    const bool second_ = k0_ >= p.c0;                                                                    \
    const int delta_ = second_ ? ((dy_ * p.a1_pitch + dx_) * p.c1 + (k0_ - p.c0)) * 6                    \
                               : ((dy_ * p.a0_pitch + dx_) * p.c0 + k0_) * 6;                            \
    unsigned char* st_ = smem_b + (ST) * C::STAGE;                                                       \
    _Pragma("unroll") for (int g = 0; g < RGA; ++g) {                                                    \
      const int iy_ = a_y[g] + dy_, ix_ = a_x[g] + dx_;                                                  \
      const bool ok_ = a_ok[g] && (unsigned)iy_ < (unsigned)p.in_h && (unsigned)ix_ < (unsigned)p.in_w;  \
      const int voff_ = ok_ ? (second_ ? a_b1[g] : a_b0[g]) + delta_ : (int)0x80000000;                  \
      unsigned char* d_ = st_ + (wave * RGA + g) * 768;                                                  \
      if (dma_lane && ABL != 3) {                                                                        \
        if (second_) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a1, (lds_void_p)(d_), 16, voff_, 0, 0, 0);    \
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a0, (lds_void_p)(d_), 16, voff_, 0, 0, 0);      \
      }                                                                                                  \
    }                                                                                                    \
    const int wslab_ = (wtap_ * p.n_total * ktot + k0_) * 6;                                             \
    _Pragma("unroll") for (int g = 0; g < CGB; ++g) {                                                    \
      unsigned char* d_ = st_ + C::A_TILE + (wave * CGB + g) * 768;                                      \
      if (dma_lane && ABL != 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void_p)(d_), 16, b_off[g], wslab_, 0, 0); \
    }                                                                                                    \
  }

  floatx16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int half = lane >> 5, l32 = lane & 31;
  // fragment read offset: row l32 of a 32-row block, half `half` of a plane (stored swapped on rows 8-15, 24-31)
  const int frag_off = l32 * C::ROWB + ((half ^ ((l32 >> 3) & 1)) * 16);

  // six bf16 products per 32x32x16 block, smallest terms first: a1 b1, a0 b2, a2 b0, a0 b1, a1 b0, a0 b0
#define ADVOC_X6D_COMPUTE(ST)                                                                            \
  {                                                                                                      \
    const unsigned char* Ax = smem_b + (ST) * C::STAGE + frag_off;                                       \
    const unsigned char* Bx = Ax + C::A_TILE;                                                            \
    bf16x8 af[MT][3], bq[NT][3];                                                                         \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                       \
      _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                   \
        af[i][pl] = *reinterpret_cast<const bf16x8*>(Ax + pl * 32 + (wm * MT + i) * 32 * C::ROWB);       \
    _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                       \
      _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                   \
        bq[j][pl] = *reinterpret_cast<const bf16x8*>(Bx + pl * 32 + (wn * NT + j) * 32 * C::ROWB);       \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                       \
      _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                   \
        if constexpr (ABL == 0 || ABL == 3) {                                                            \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bq[j][1], acc[i][j], 0, 0, 0);     \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bq[j][2], acc[i][j], 0, 0, 0);     \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bq[j][0], acc[i][j], 0, 0, 0);     \
        }                                                                                                \
        if constexpr (ABL != 2) {                                                                        \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bq[j][1], acc[i][j], 0, 0, 0);     \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bq[j][0], acc[i][j], 0, 0, 0);     \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bq[j][0], acc[i][j], 0, 0, 0);     \
        } else {                                                                                         \
          asm volatile("" ::"v"(af[i][0]), "v"(af[i][1]), "v"(af[i][2]), "v"(bq[j][0]), "v"(bq[j][1]), "v"(bq[j][2])); \
        }                                                                                                \
      }                                                                                                  \
  }

  // ---- K loop: NS stages, one barrier per K tile ----
  // prologue: tiles kt_begin .. kt_begin + NS - 2 into stages 0 .. NS - 2
  ADVOC_X6D_ISSUE(0);
  if constexpr (NS == 3) ADVOC_X6D_ISSUE(1);

  int kt = kt_begin;
  // NS iterations per trip so that stage indices are compile-time constants; a trip may overshoot kt_end by
  // up to NS - 1 tiles (wrapped loads above, compute skipped below)
  while (kt < kt_end) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      // tile kt + u sits in stage u; the next tile to issue goes into stage (u + NS - 1) % NS, vacated by tile kt + u - 1
      wait_vmcnt<C::DMA_PER_TILE*(NS - 2)>();
      __builtin_amdgcn_s_barrier();
      ADVOC_X6D_ISSUE((u + NS - 1) % NS);
      if (ABL != 4 && kt + u < kt_end) ADVOC_X6D_COMPUTE(u);
    }
    kt += NS;
  }
#undef ADVOC_X6D_ISSUE
#undef ADVOC_X6D_COMPUTE
  // the overshoot DMAs still target LDS: drain them before the epilogue reuses it
  wait_vmcnt<0>();
  __syncthreads();

  // ---- tail slices (igemm.hip): park the partial tile; the LAST slice to arrive sums all of them in slice
  // order and carries on into the ordinary epilogue ----
  if (tail_tile >= 0) {
    constexpr int TILE = BM * BN;
    float* part = p.tail_ws + ((size_t)tail_tile * ks_cnt + ks_idx) * TILE;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          __hip_atomic_store(part + (((wave * MT + i) * NT + j) * 16 + r) * 64 + lane, acc[i][j][r],
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);        // this thread's partial has reached the coherent level ...
    __syncthreads();
    int* s_flag = reinterpret_cast<int*>(smem);
    if (tid == 0)                         // ... before the workgroup is counted
      *s_flag = __hip_atomic_fetch_add(p.tail_cnt + tail_tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int arrived = *s_flag;
    __syncthreads();                      // smem is reused below
    if (arrived != ks_cnt - 1) return;
    if (tid == 0)                         // ready for the next launch
      __hip_atomic_store(p.tail_cnt + tail_tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float* all = p.tail_ws + (size_t)tail_tile * ks_cnt * TILE;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        floatx16 sum;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum[r] = 0.f;
        for (int sl = 0; sl < ks_cnt; ++sl) {
          const float* src = all + (size_t)sl * TILE + (((wave * MT + i) * NT + j) * 16) * 64 + lane;
#pragma unroll
          for (int r = 0; r < 16; ++r)
            sum[r] += __hip_atomic_load(src + r * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        acc[i][j] = sum;
      }
  }
  const bool atomic_split = tail_tile < 0 && ks_cnt > 1;

  // ---- epilogue (igemm.hip): pixel table, LDS transpose, 16-byte stores with the fused bias / dropout /
  // activation-gradient / two-destination logic ----
  for (int r = tid; r < BM; r += 256) {
    const unsigned m = m0 + r;
    int pix0 = -1, pix1 = -1;
    if (m < M) {
      const unsigned t = m / (unsigned)p.gw;
      const int gx = (int)(m - t * (unsigned)p.gw);
      const int img = (int)(t / (unsigned)p.gh);
      const int gy = (int)(t - (unsigned)img * (unsigned)p.gh);
      const int oy = gy * p.osy + p.ooy[phase], ox = gx * p.osx + p.oox[phase];
      if (oy < p.out_h && ox < p.out_w) {
        pix0 = (img * p.out_h + oy) * p.d[0].pitch + ox;
        pix1 = (img * p.out_h + oy) * p.d[1].pitch + ox;
      }
    }
    s_pix[r] = pix0;
    s_pix[BM + r] = pix1;
  }
  __syncthreads();

  const float gslope = act_slope(p.grad_act);
  constexpr int LDT = 36;
  float* T = smem + wave * (32 * LDT);
  const int trow = lane >> 3, tq = lane & 7;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int nt0 = n0 + (wn * NT + j) * 32;
    const int di = nt0 >= p.n_split ? 1 : 0;
    const GemmDest& d = p.d[di];
    if (d.p == nullptr) continue;
    const int ch = (di ? nt0 - p.n_split : nt0) + 4 * tq;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && (ks_idx == 0 || !atomic_split)) bias4 = *reinterpret_cast<const float4*>(p.bias + nt0 + 4 * tq);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * half) * LDT + l32] = acc[i][j][r];
      wave_lds_sync();
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int row = trow + 8 * ps;
        const int pix = s_pix[di * BM + (wm * MT + i) * 32 + row];
        if (pix < 0) continue;
        const int off = pix * d.c + ch;
        float4 v = *reinterpret_cast<const float4*>(T + row * LDT + 4 * tq);
        v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
        if (p.y_mask) {
          const uchar4 mk = *reinterpret_cast<const uchar4*>(p.y_mask + off);
          v.x *= mk.x * p.y_mask_scale; v.y *= mk.y * p.y_mask_scale;
          v.z *= mk.z * p.y_mask_scale; v.w *= mk.w * p.y_mask_scale;
        }
        if (p.grad_act != ADVOC_ACT_NONE) {
          float4 x = *reinterpret_cast<const float4*>(d.xpre + off);
          if (d.gscale) {
            const float4 gs = *reinterpret_cast<const float4*>(d.gscale + ch);
            const float4 gh = *reinterpret_cast<const float4*>(d.gshift + ch);
            x.x = x.x * gs.x + gh.x; x.y = x.y * gs.y + gh.y; x.z = x.z * gs.z + gh.z; x.w = x.w * gs.w + gh.w;
          }
          v.x *= x.x > 0.f ? 1.f : gslope; v.y *= x.y > 0.f ? 1.f : gslope;
          v.z *= x.z > 0.f ? 1.f : gslope; v.w *= x.w > 0.f ? 1.f : gslope;
        }
        if (d.gmask) {
          const uchar4 mk = *reinterpret_cast<const uchar4*>(d.gmask + off);
          v.x *= mk.x * d.gmask_scale; v.y *= mk.y * d.gmask_scale;
          v.z *= mk.z * d.gmask_scale; v.w *= mk.w * d.gmask_scale;
        }
        if (atomic_split) {
          unsafeAtomicAdd(d.p + off, v.x); unsafeAtomicAdd(d.p + off + 1, v.y);
          unsafeAtomicAdd(d.p + off + 2, v.z); unsafeAtomicAdd(d.p + off + 3, v.w);
          continue;
        }
        if (d.accum) {
          const float4 o = *reinterpret_cast<const float4*>(d.p + off);
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *reinterpret_cast<float4*>(d.p + off) = v;
      }
      wave_lds_sync();
    }
  }
}

template <int MT, int NT, int NS>
__global__ __launch_bounds__(256, 2) void gather_gemm_x6d_kernel(const GatherGemmParams p) {
  gather_gemm_x6d_body<MT, NT, NS>(p);
}
// timing experiments only (ADVOC_X6D_ABLATE=1..4): 1 three of the six products, 2 no MFMAs (fragment reads kept),
// 3 no DMA, 4 DMA only
template <int MT, int NT, int NS, int ABL>
__global__ __launch_bounds__(256, 2) void gather_gemm_x6d_abl_kernel(const GatherGemmParams p) {
  gather_gemm_x6d_body<MT, NT, NS, ABL>(p);
}

template <int MT, int NT, int NS>
int launch_d(const GatherGemmParams& p, hipStream_t stream, const char** name_only, const TailPlan& tail,
             float* tail_ws, int* tail_cnt) {
  using C = DCfg<MT, NT, NS>;
  if (name_only) {
    static const std::string name = std::string("gather_gemm_x6d_kernel<") + std::to_string(MT) + ", " +
                                    std::to_string(NT) + ", " + std::to_string(NS) + ">";
    *name_only = name.c_str();
    return ADVOC_OK;
  }
  static const int abl = getenv("ADVOC_X6D_ABLATE") ? atoi(getenv("ADVOC_X6D_ABLATE")) : 0;
  auto kern = gather_gemm_x6d_kernel<MT, NT, NS>;
  if (NS == 2 && NT == 4) {   // experiments: one instance is enough
    if (abl == 1) kern = gather_gemm_x6d_abl_kernel<MT, NT, NS, 1>;
    if (abl == 2) kern = gather_gemm_x6d_abl_kernel<MT, NT, NS, 2>;
    if (abl == 3) kern = gather_gemm_x6d_abl_kernel<MT, NT, NS, 3>;
    if (abl == 4) kern = gather_gemm_x6d_abl_kernel<MT, NT, NS, 4>;
  }
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
  if (attr != hipSuccess) { note_hip_error(attr); return ADVOC_ERR_HIP; }
  const int64_t M = (int64_t)p.batch * p.gh * p.gw;
  const int64_t gx = ceil_div(M, C::BM) * (p.n_total / C::BN);
  GatherGemmParams q = p;
  dim3 grid((unsigned)gx, 1, (unsigned)p.nphase);
  if (tail.split > 1 && tail_ws && tail_cnt) {
    q.tail_main = tail.main; q.tail_split = tail.split; q.tail_ws = tail_ws; q.tail_cnt = tail_cnt;
    grid = dim3((unsigned)(tail.main + tail.rem * tail.split), 1, 1);
  }
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, stream, q);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

struct Pick { int mt, nt, ns; };

// Tile choice.  128 x 256 / 256 x 128 halve the L2 -> LDS traffic of one operand per flop; they need >= ~2 tiles
// per CU to fill the chip.  Measured on MI355X (tools/layer_times.py), see DESIGN.md §4.
Pick pick_tile(const GatherGemmParams& p) {
  const Tuning& t = tuning();
  const int N = p.n_total;
  const int64_t rows128 = ceil_div((int64_t)p.batch * p.gh * p.gw, 128) * p.nphase;
  Pick k = {2, 2, 3};
  if (t.x6d_tile == 2 && N % 256 == 0) k = {2, 4, 2};
  else if (t.x6d_tile == 3) k = {4, 2, 2};
  else if (t.x6d_tile == 0) {
    if (N % 256 == 0 && rows128 * (N / 256) >= 512) k = {2, 4, 2};
  }
  if (t.x6d_stages == 2 || t.x6d_stages == 3) k.ns = t.x6d_stages;
  return k;
}

}  // namespace

// Workspace layout of one launch: [weight image][image of source 0][image of source 1][tail partials]
int64_t x6d_round(int64_t b) { return (b + 255) / 256 * 256; }

bool x6d_eligible(const GatherGemmParams& p) {
  const Tuning& t = tuning();
  if (!t.x6d || !t.igemm_x6) return false;
  const int ktot = p.c0 + p.c1, N = p.n_total;
  if (ktot % 16 || p.c0 % 16 || p.c1 % 16 || N % 128 || p.n_split % 32) return false;
  if (p.n_valid && p.n_valid != p.n_total) return false;
  const int64_t rows128 = ceil_div((int64_t)p.batch * p.gh * p.gw, 128) * p.nphase;
  if (rows128 * (N / 128) < t.x6d_min_tiles) return false;
  // 32-bit byte offsets inside the kernel: three planes of either source, and of the weights, below 2 GiB
  const int64_t lim = 0x7fffffffLL;
  const int64_t e0 = (int64_t)p.batch * p.a_h * p.a0_pitch * p.c0, e1 = (int64_t)p.batch * p.a_h * p.a1_pitch * p.c1;
  if (6 * e0 > lim || 6 * e1 > lim) return false;
  int taps = 0;
  for (int ph = 0; ph < p.nphase; ++ph)
    for (int i = 0; i < p.ntaps; ++i) taps = (p.tap[ph][i] >> 16) + 1 > taps ? (p.tap[ph][i] >> 16) + 1 : taps;
  if ((int64_t)6 * taps * N * ktot > lim) return false;
  return true;
}

int launch_gather_gemm_x6d(const GatherGemmParams& p_in, bool b_kn, hipStream_t stream, const char** name_only,
                           float* scratch, int64_t scratch_bytes, int64_t* scratch_query) {
  if (!x6d_eligible(p_in)) return ADVOC_ERR_UNSUPPORTED;
  GatherGemmParams p = p_in;
  const int ktot = p.c0 + p.c1, N = p.n_total;
  int taps = 0;
  for (int ph = 0; ph < p.nphase; ++ph)
    for (int i = 0; i < p.ntaps; ++i) taps = (p.tap[ph][i] >> 16) + 1 > taps ? (p.tap[ph][i] >> 16) + 1 : taps;
  const int64_t e0 = (int64_t)p.batch * p.a_h * p.a0_pitch * p.c0, e1 = (int64_t)p.batch * p.a_h * p.a1_pitch * p.c1;
  const int64_t wq_bytes = x6d_round((int64_t)6 * taps * N * ktot);
  const int64_t i0_bytes = x6d_round(6 * e0), i1_bytes = x6d_round(6 * e1);
  const Pick k = pick_tile(p);
  const int BM = 64 * k.mt, BN = 64 * k.nt;
  const int64_t tiles = ceil_div((int64_t)p.batch * p.gh * p.gw, BM) * (N / BN) * p.nphase;
  const int nkt = ktot / 16 * p.ntaps;
  const TailPlan tail = plan_tail(tiles, nkt);
  const int64_t tail_bytes = (int64_t)sizeof(float) * tail.rem * tail.split * BM * BN;
  const int64_t need = wq_bytes + i0_bytes + i1_bytes;
  if (scratch_query) { *scratch_query = need + tail_bytes; return ADVOC_OK; }
  if (!scratch || scratch_bytes < need) return ADVOC_ERR_UNSUPPORTED;
  p.k_order = tuning().igemm_korder >= 0 ? tuning().igemm_korder : 1;
  char* ws = reinterpret_cast<char*>(scratch);
  p.wq = reinterpret_cast<const uint16_t*>(ws);
  p.wq_taps = taps;
  p.a0_img = reinterpret_cast<const uint16_t*>(ws + wq_bytes);
  p.a1_img = e1 ? reinterpret_cast<const uint16_t*>(ws + wq_bytes + i0_bytes) : nullptr;
  p.a0_img_bytes = (int)(6 * e0);
  p.a1_img_bytes = (int)(6 * e1);
  float* tail_ws = scratch_bytes >= need + tail_bytes ? reinterpret_cast<float*>(ws + need) : nullptr;
  int* tail_cnt = nullptr;
  if (!name_only && tuning().x6d_skip_prep) {
    if (tail.split > 1 && tail_ws) tail_cnt = tail_counter_slot();
  } else if (!name_only) {
    int rc = launch_split_weights(p.w, reinterpret_cast<uint16_t*>(ws), taps, N, N, ktot, b_kn, true, stream);
    if (rc != ADVOC_OK) return rc;
    rc = launch_split_image(p.a0, reinterpret_cast<uint16_t*>(ws + wq_bytes), e0, p.c0, p.in_scale, p.in_shift,
                            p.in_act, p.a_mask, p.a_mask_scale, stream);
    if (rc != ADVOC_OK) return rc;
    if (e1) {
      rc = launch_split_image(p.a1, reinterpret_cast<uint16_t*>(ws + wq_bytes + i0_bytes), e1, p.c1,
                              p.in_scale ? p.in_scale + p.c0 : nullptr, p.in_shift ? p.in_shift + p.c0 : nullptr,
                              p.in_act, nullptr, 0.f, stream);
      if (rc != ADVOC_OK) return rc;
    }
    if (tail.split > 1 && tail_ws) tail_cnt = tail_counter_slot();
  }
  if (k.mt == 2 && k.nt == 2 && k.ns == 3) return launch_d<2, 2, 3>(p, stream, name_only, tail, tail_ws, tail_cnt);
  if (k.mt == 2 && k.nt == 2) return launch_d<2, 2, 2>(p, stream, name_only, tail, tail_ws, tail_cnt);
  if (k.mt == 2 && k.nt == 4 && k.ns == 3) return launch_d<2, 4, 3>(p, stream, name_only, tail, tail_ws, tail_cnt);
  if (k.mt == 2 && k.nt == 4) return launch_d<2, 4, 2>(p, stream, name_only, tail, tail_ws, tail_cnt);
  if (k.mt == 4 && k.nt == 2 && k.ns == 3) return launch_d<4, 2, 3>(p, stream, name_only, tail, tail_ws, tail_cnt);
  return launch_d<4, 2, 2>(p, stream, name_only, tail, tail_ws, tail_cnt);
}

}  // namespace advoc
