// Gather-GEMM on operand images: the convolution kernel whose operands arrive as fp16 PAIRS, one cache line
// per tile row.
//
// Same GEMM view as igemm.hip (rows = output grid points, cols = output channels, depth = (tap, input
// channel); replaces the cuDNN / Eigen Conv2D and Conv2DBackpropInput kernels TF1 runs for
// models/advoc/advoc_model.py:25-69 in both directions).  Arithmetic (image.hip): every fp32 operand, scaled by
// one power of two per tensor, is the sum of two fp16 numbers to 2^-22; a product is three fp16 MFMA products
// a0 b1 + a1 b0 + a0 b0 with fp32 accumulation, unscaled exactly in the epilogue -- fp32-level error at 3 / 16
// of the fp32 MFMA cost (roof in algorithmic fp32 flops: dense f16 MFMA / 3).
//
// Why it looks the way it does (measured on MI355X, NOTEBOOK.md §4): the register-split kernel of igemm.hip and a
// first image kernel (bf16 triples, six products) both stopped at ~170-190 algorithmic TFLOP/s with the matrix
// pipe busy 45 % of the time; removing every MFMA from the loop bought 16 %, removing the loads 49 %.  The loaders
// were bound by the L2's request rate: 32-96 useful bytes of every 128-byte line they asked for.  Here
//   * one row of a K tile (32 contraction slots x 2 planes) is exactly ONE 128-byte line of the image, fetched once;
//   * both operands go global -> LDS by `buffer_load_dwordx4 ... lds` (LDS-DMA): no VGPR staging, no ds_write, no
//     conversion.  The hardware range check of the buffer descriptor supplies the zero padding: a tap that falls
//     outside the input gets an offset beyond num_records and the DMA writes zeros;
//   * one DMA instruction = 8 tile rows x 8 sixteen-byte chunks (1 KiB of LDS, all 64 lanes); a wave issues BM / 32
//     of them per K tile for A and BN / 32 for B, each lane keeping ONE (row, chunk) per instruction slot for the
//     whole K loop: per K tile the address work is two compares and a select per slot, everything else is scalar;
//   * LDS rows are 128 bytes; chunk c of row r sits at position c ^ ((r >> 1) & 7) (applied to the SOURCE address,
//     the DMA destination is lane-linear), which makes the MFMA fragment reads (ds_read_b128, one row per lane,
//     16 rows per LDS pass) conflict-free;
//   * NS LDS stages (2 or 3), ONE s_barrier per K tile: wait for the own DMAs of tile t (counted vmcnt, the
//     younger tiles stay in flight) -> barrier -> issue tile t + NS - 1 into the stage tile t - 1 just
//     vacated -> 16 ds_read_b128 + 24 MFMAs (128 x 128 tile) of tile t.
// Tile mapping (XCD-aware order), split-K, the tail split and the epilogue are those of igemm.hip.
#include <stdlib.h>

#include <cstdio>
#include <string>

#include "common.h"
#include "igemm.h"
#include "tuning.h"
#include "image_emit.h"
#include "lds_dma.h"
#include "x6.h"

namespace advoc {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_void_p;

__device__ __forceinline__ float act_slope(int act) {
  return act == ADVOC_ACT_LRELU02 ? 0.2f : (act == ADVOC_ACT_RELU ? 0.f : 1.f);
}

// WGM x 2 wavefronts per workgroup (WGM = 2: 256 threads; WGM = 4: 512 threads, one workgroup per CU with 2-stage tiles of
// 256 rows -- twice the rows per byte of B streamed)
template <int MT, int NT, int NS, int WGM_ = 2>
struct HCfg {
  static constexpr int WGM = WGM_, WGN = 2;
  static constexpr int WAVES = WGM * WGN, THREADS = 64 * WAVES;
  static constexpr int BM = 32 * MT * WGM, BN = 32 * NT * WGN;
  static constexpr int BK = 32;
  static constexpr int ROWB = 4 * BK;                       // bytes of one tile row: 2 planes x 32 fp16 = one line
  static constexpr int A_TILE = BM * ROWB, B_TILE = BN * ROWB;
  static constexpr int STAGE = A_TILE + B_TILE;             // bytes
  static constexpr int RGA = BM / (8 * WAVES), CGB = BN / (8 * WAVES);   // 8-row DMA blocks per wave and K tile
  static constexpr int DMA_PER_TILE = RGA + CGB;
  static constexpr int EPI_BYTES = WAVES * 32 * 36 * 4 + 2 * BM * 4;
  static_assert(BM % (8 * WAVES) == 0 && BN % (8 * WAVES) == 0, "whole DMA blocks per wave");
  static constexpr size_t LDS_BYTES = NS * STAGE > EPI_BYTES ? NS * STAGE : EPI_BYTES;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// (the body is a __device__ function behind a one-line kernel: with the body inside the __global__ template itself
// hipcc's HOST pass silently dropped the kernel handle -- no diagnostic, an undefined symbol at load time)
template <int MT, int NT, int NS, int WGM, int ABL = 0>
__device__ __forceinline__ void gather_gemm_h3_body(const GatherGemmParams& p) {
  using C = HCfg<MT, NT, NS, WGM>;
  constexpr int BM = C::BM, BN = C::BN, BK = C::BK, WGN = C::WGN;
  constexpr int RGA = C::RGA, CGB = C::CGB;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* smem_b = reinterpret_cast<unsigned char*>(smem);
  int* s_pix = reinterpret_cast<int*>(smem + C::WAVES * 32 * 36);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const unsigned M = (unsigned)p.batch * (unsigned)p.gh * (unsigned)p.gw;
  const int ntn = (p.n_total + BN - 1) / BN;       // (n_total = 32: one 64-column tile whose upper half is masked, below)
  int tile, phase, ks_idx = blockIdx.y, ks_cnt = gridDim.y, tail_tile = -1;
  {
    const bool tail_mode = p.tail_split > 1;       // tail_main == 0: EVERY tile is cut into K slices
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

  // ---- DMA lanes: lane -> (row lane / 8 of an 8-row block, LDS position lane % 8 of the 128-byte row); position q of
  // row r holds chunk q ^ ((r >> 1) & 7) of the line ----
  const int lrow = lane >> 3, lpos = lane & 7;
  int a_y[RGA], a_x[RGA], a_b0[RGA], a_b1[RGA];
  bool a_ok[RGA];
#pragma unroll
  for (int g = 0; g < RGA; ++g) {
    const int r = (wave * RGA + g) * 8 + lrow;
    const unsigned m = m0 + r;
    a_ok[g] = m < M;
    const unsigned mm = a_ok[g] ? m : 0u;
    const unsigned t = mm / (unsigned)p.gw;
    const int gx = (int)(mm - t * (unsigned)p.gw);
    const int img = (int)(t / (unsigned)p.gh);
    const int gy = (int)(t - (unsigned)img * (unsigned)p.gh);
    a_x[g] = (gx + p.gx_off) * p.sx;
    a_y[g] = gy * p.sy;
    const int gc = lpos ^ ((r >> 1) & 7);
    a_b0[g] = (((img * p.a_h + a_y[g]) * p.a0_pitch + a_x[g]) * p.c0) * 4 + gc * 16;     // bytes into the image
    a_b1[g] = (((img * p.a_h + a_y[g]) * p.a1_pitch + a_x[g]) * p.c1) * 4 + gc * 16;
    a_b1[g] -= a_b0[g];        // kept as the DIFFERENCE: `second ? a_b1[g] : a_b0[g]` in the K loop made the compiler merge
                               // the two arrays into one in scratch memory, indexed by `second` -- a scratch load and
                               // an s_waitcnt vmcnt(0), i.e. a wait for every DMA in flight, per DMA slot and K tile
  }
  int b_off[CGB];
#pragma unroll
  for (int g = 0; g < CGB; ++g) {
    const int n = (wave * CGB + g) * 8 + lrow;
    const int gc = lpos ^ ((n >> 1) & 7);
    // (rows beyond the weight image -- the 32-channel layers of AdVoc-small on the 64-column tile -- arrive as zeros:
    // behind the last tap the descriptor's range check would do it, behind the others they would alias the next tap)
    b_off[g] = n0 + n < p.n_total ? ((n0 + n) * ktot) * 4 + gc * 16 : (int)0x80000000;
  }
  // buffer descriptors over the whole images; the weight-slab / K-slice offsets go into soffset
  // (lds_dma.h: why the DMAs are inline assembly)
  const u32x4s rs_a0 = dma_rsrc(p.a0_img, (unsigned)p.a0_img_bytes);
  const u32x4s rs_a1 = dma_rsrc(p.a1_img ? p.a1_img : p.a0_img, (unsigned)p.a1_img_bytes);
  const u32x4s rs_b = dma_rsrc(p.wq, (unsigned)(p.wq_taps * p.n_total * ktot * 4));
  const unsigned lds0 = lds_address(smem_b);

  const int kt_begin = (int)((int64_t)nkt * ks_idx / ks_cnt);
  const int kt_end = (int)((int64_t)nkt * (ks_idx + 1) / ks_cnt);
  const bool taps_inner = p.k_order != 0;
  int ld_tap = taps_inner ? kt_begin % p.ntaps : kt_begin / kpt;
  int ld_k0 = (taps_inner ? kt_begin / p.ntaps : kt_begin % kpt) * BK;

  // the tap table of this phase, one entry per lane, read with v_readlane in the K loop: an s_load there is followed by
  // s_waitcnt lgkmcnt(0), i.e. by the scalar memory latency once per K tile
  const int tapv = p.tap[phase][lane < p.ntaps ? lane : 0];
  // Issues the DMAs of the next K tile of the walk into stage ST (a compile-time constant in the unrolled
  // loop below).  Tiles past kt_end wrap round to valid ones; their data is never read.
#define ADVOC_H3_ISSUE(ST)                                                                               \
  {                                                                                                      \
    const int ti_ = ld_tap;                                                                              \
    const int k0_ = ld_k0;                                                                               \
    if (taps_inner) {                                                                                    \
      if (++ld_tap == p.ntaps) { ld_tap = 0; ld_k0 += BK; if (ld_k0 == ktot) ld_k0 = 0; }                \
    } else {                                                                                             \
      ld_k0 += BK;                                                                                       \
      if (ld_k0 == ktot) { ld_k0 = 0; if (++ld_tap == p.ntaps) ld_tap = 0; }                             \
    }                                                                                                    \
    const int tp_ = __builtin_amdgcn_readlane(tapv, ti_);                                                \
    const int dy_ = (int)(int8_t)(tp_ & 0xff), dx_ = (int)(int8_t)((tp_ >> 8) & 0xff);                   \
    const int wtap_ = tp_ >> 16;                                                                         \
    const bool second_ = k0_ >= p.c0;                                                                    \
    const int delta_ = second_ ? ((dy_ * p.a1_pitch + dx_) * p.c1 + (k0_ - p.c0)) * 4                    \
                               : ((dy_ * p.a0_pitch + dx_) * p.c0 + k0_) * 4;                            \
    const unsigned st_ = lds0 + (ST) * C::STAGE;                                                         \
    const u32x4s rs_a_ = second_ ? rs_a1 : rs_a0;                                                        \
    _Pragma("unroll") for (int g = 0; g < RGA; ++g) {                                                    \
      const int iy_ = a_y[g] + dy_, ix_ = a_x[g] + dx_;                                                  \
      const bool ok_ = a_ok[g] && (unsigned)iy_ < (unsigned)p.in_h && (unsigned)ix_ < (unsigned)p.in_w;  \
      const int voff_ = ok_ ? a_b0[g] + (second_ ? a_b1[g] : 0) + delta_ : (int)0x80000000;              \
      if (ABL != 3) dma16(rs_a_, st_ + (wave * RGA + g) * 1024, voff_);                                  \
    }                                                                                                    \
    const int wslab_ = (wtap_ * p.n_total * ktot + k0_) * 4;                                             \
    _Pragma("unroll") for (int g = 0; g < CGB; ++g) {                                                    \
      if (ABL != 3) dma16(rs_b, st_ + C::A_TILE + (wave * CGB + g) * 1024, b_off[g], wslab_);            \
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
  // fragment read offsets: row l32 of a 32-row block; chunk (plane, k step, half) = 4 plane + 2 ks + half sits at
  // position chunk ^ ((row >> 1) & 7)
  int frag_off[2][2];
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      frag_off[pl][ks] = l32 * C::ROWB + (((4 * pl + 2 * ks + half) ^ ((l32 >> 1) & 7)) * 16);

  // three fp16 products per 32x32x16 block, small terms first: a0 b1, a1 b0, a0 b0
#define ADVOC_H3_COMPUTE(ST)                                                                             \
  {                                                                                                      \
    const unsigned char* Ax = smem_b + (ST) * C::STAGE;                                                  \
    const unsigned char* Bx = Ax + C::A_TILE;                                                            \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                   \
      f16x8 af[MT][2], bq[NT][2];                                                                        \
      _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                     \
        _Pragma("unroll") for (int pl = 0; pl < 2; ++pl)                                                 \
          af[i][pl] = *reinterpret_cast<const f16x8*>(Ax + frag_off[pl][ks] + (wm * MT + i) * 32 * C::ROWB); \
      _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                     \
        _Pragma("unroll") for (int pl = 0; pl < 2; ++pl)                                                 \
          bq[j][pl] = *reinterpret_cast<const f16x8*>(Bx + frag_off[pl][ks] + (wn * NT + j) * 32 * C::ROWB); \
      _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                     \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                 \
          if constexpr (ABL != 2) {                                                                      \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][0], bq[j][1], acc[i][j], 0, 0, 0);    \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][1], bq[j][0], acc[i][j], 0, 0, 0);    \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][0], bq[j][0], acc[i][j], 0, 0, 0);    \
          } else {                                                                                       \
            asm volatile("" ::"v"(af[i][0]), "v"(af[i][1]), "v"(bq[j][0]), "v"(bq[j][1]));               \
          }                                                                                              \
        }                                                                                                \
    }                                                                                                    \
  }

  // ---- K loop: NS stages, one barrier per K tile ----
  // prologue: tiles kt_begin .. kt_begin + NS - 2 into stages 0 .. NS - 2
  ADVOC_H3_ISSUE(0);
  if constexpr (NS >= 3) ADVOC_H3_ISSUE(1);
  if constexpr (NS >= 4) ADVOC_H3_ISSUE(2);

  int kt = kt_begin;
  // NS iterations per trip so that stage indices are compile-time constants; a trip may overshoot kt_end by
  // up to NS - 1 tiles (wrapped loads above, compute skipped below)
  while (kt < kt_end) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      // tile kt + u sits in stage u; the next tile to issue goes into stage (u + NS - 1) % NS, vacated by tile kt + u - 1
      // (lds_dma.h: the rendezvous also waits for this wave's own fragment reads of the stage the DMAs below overwrite)
#ifdef ADVOC_H3_IGLP
      __builtin_amdgcn_iglp_opt(ADVOC_H3_IGLP);
#endif
      dma_ring_barrier<C::DMA_PER_TILE*(NS - 2)>();
      ADVOC_H3_ISSUE((u + NS - 1) % NS);
      if (ABL != 4 && kt + u < kt_end) ADVOC_H3_COMPUTE(u);
    }
    kt += NS;
  }
#undef ADVOC_H3_ISSUE
#undef ADVOC_H3_COMPUTE
  // the overshoot DMAs still target LDS: drain them before the epilogue reuses it
  wait_vmcnt<0>();
  __syncthreads();

  // ---- tail slices (igemm.hip): park the partial tile; the LAST slice to arrive sums all of them in slice
  // order and carries on into the ordinary epilogue ----
  if (tail_tile >= 0) {
    // A partial tile is parked block by block (32 x 32 accumulators of one wave), the lane's 16 values as four 16-byte
    // pieces: piece q of every lane is 1 KB contiguous.  All accesses are agent-scope (sc1: past the XCD's own L2, which
    // is not coherent with the others'), as the relaxed agent-scope atomics they replace were -- but 16 bytes per lane
    // instead of 4, and the reader keeps FOUR slices of a block in flight instead of one: the last slice to arrive used
    // to pay one memory round trip (> 1 us at this scope) per slice and block, 32 of them in a row for a 16-slice tile
    // (r4: 65 us for the 1.6 GFLOP of encoder_8, two thirds of it here).
    constexpr int TILE = BM * BN;
    constexpr int kAgent = 16;                          // cache policy bit sc1
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(p.tail_ws, 0, 0xffffff00u, 0x00020000);
    const unsigned tile_b = (unsigned)tail_tile * (unsigned)ks_cnt * (unsigned)(TILE * 4);    // (host: the workspace < 4 GB)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          u32x4 v;
          v.x = __float_as_uint(acc[i][j][4 * q]); v.y = __float_as_uint(acc[i][j][4 * q + 1]);
          v.z = __float_as_uint(acc[i][j][4 * q + 2]); v.w = __float_as_uint(acc[i][j][4 * q + 3]);
          __builtin_amdgcn_raw_buffer_store_b128(v, rs_t, tile_b + ((((wave * MT + i) * NT + j) * 4 + q) * 64 + lane) * 16,
                                                 ks_idx * (TILE * 4), kAgent);
        }
    // (r3: an agent-scope RELEASE fence here -- buffer_wbl2, a write-back of this XCD's whole L2 -- was tried while chasing
    // a side-stream race and cost +46 % on this kernel (58 -> 85 us average, +1 ms per step) without changing the race:
    // the partial tiles are written with agent-scope stores that bypass the non-coherent L2 path already.)
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
    // the sum runs in slice order whatever the order of arrival: results do not depend on the schedule
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        floatx16 sum;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum[r] = 0.f;
        const unsigned blk = tile_b + ((((wave * MT + i) * NT + j) * 4) * 64 + lane) * 16;
        int sl = 0;
        for (; sl + 4 <= ks_cnt; sl += 4) {
          u32x4 t[4][4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              t[u][q] = __builtin_amdgcn_raw_buffer_load_b128(rs_t, blk + q * 1024, (sl + u) * (TILE * 4), kAgent);
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              sum[4 * q] += __uint_as_float(t[u][q].x); sum[4 * q + 1] += __uint_as_float(t[u][q].y);
              sum[4 * q + 2] += __uint_as_float(t[u][q].z); sum[4 * q + 3] += __uint_as_float(t[u][q].w);
            }
        }
        for (; sl < ks_cnt; ++sl) {
          u32x4 t[4];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            t[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_t, blk + q * 1024, sl * (TILE * 4), kAgent);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            sum[4 * q] += __uint_as_float(t[q].x); sum[4 * q + 1] += __uint_as_float(t[q].y);
            sum[4 * q + 2] += __uint_as_float(t[q].z); sum[4 * q + 3] += __uint_as_float(t[q].w);
          }
        }
        acc[i][j] = sum;
      }
  }
  const bool atomic_split = tail_tile < 0 && ks_cnt > 1;

  // ---- epilogue (igemm.hip): pixel table, LDS transpose, 16-byte stores with the fused bias / dropout /
  // activation-gradient / two-destination logic ----
  for (int r = tid; r < BM; r += C::THREADS) {
    const unsigned m = m0 + r;
    int pix0 = -1, pix1 = -1;
    if (m < M) {
      const unsigned t = m / (unsigned)p.gw;
      const int gx = (int)(m - t * (unsigned)p.gw);
      const int img = (int)(t / (unsigned)p.gh);
      const int gy = (int)(t - (unsigned)img * (unsigned)p.gh);
      const int oy = gy * p.osy + p.ooy[phase], ox = (gx + p.gx_off) * p.osx + p.oox[phase];
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
  // undo the operands' power-of-two scaling (exact)
  const float unscale = __uint_as_float(p.a_hdr[1]) * __uint_as_float(p.b_hdr[1]);
  constexpr int LDT = 36;
  float* T = smem + wave * (32 * LDT);
  const int trow = lane >> 3, tq = lane & 7;
  // consumers' operand images of the output (image_emit.h): one-pass scale from the consumer header's previous magnitude
  const bool emit0 = p.oimg[0].img != nullptr && !atomic_split, emit1 = p.oimg[1].img != nullptr && !atomic_split;
  // (r5) oimg_bounded: the a-priori scale of igemm_patch.hip (this kernel takes the remainder columns of a patch launch and
  // the launches under 16 x 16 grid points: both write the same image under the same bound)
  float eup0 = 1.f;
  if (emit0) {
    if (p.oimg_bounded) {
      float bound = __uint_as_float(*p.a_amax) * emit_weight_bound(p, ktot);
      if (p.obound_add) bound += __uint_as_float(*p.obound_add);
      if (p.grad_act == ADVOC_ACT_NONE && p.bias) {      // a forward launch: + max |b|
        float bm = 0.f;
        for (int n = lane; n < p.n_total; n += 64) bm = fmaxf(bm, fabsf(p.bias[n]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) bm = fmaxf(bm, __shfl_xor(bm, off, 64));
        bound += bm;
      }
      eup0 = emit_up_scale_bounded(bound);
    } else {
      eup0 = emit_up_scale(p.oimg[0].hdr[2]);
    }
  }
  const float eup1 = emit1 ? emit_up_scale(p.oimg[1].hdr[2]) : 1.f;
  float evmax0 = 0.f, evmax1 = 0.f, dmax1 = 0.f;
  if (tid == 0) {      // (every workgroup that reaches an epilogue: the same value)
    if (emit0) p.oimg[0].hdr[1] = __float_as_uint(1.f / eup0);
    if (emit1) p.oimg[1].hdr[1] = __float_as_uint(1.f / eup1);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int nt0 = n0 + (wn * NT + j) * 32;
    if (nt0 >= p.n_total) continue;                 // (masked half of a 64-column tile over 32 channels)
    const int di = nt0 >= p.n_split ? 1 : 0;
    const GemmDest& d = p.d[di];
    if (d.p == nullptr) continue;
    const int ch = (di ? nt0 - p.n_split : nt0) + 4 * tq;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && (ks_idx == 0 || !atomic_split)) bias4 = *reinterpret_cast<const float4*>(p.bias + nt0 + 4 * tq);
    // per-channel sums of destination 0 over the tile's rows (ocolsum_table: the bias gradient of the layer whose
    // output-gradient image this launch writes)
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool no_store = p.d0_no_store && di == 0 && emit0;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      // ALL global loads of the block (pre-activation values, masks, the value to add to) are issued before anything is
      // used or stored: loads and stores retire through one in-order counter, so a load placed after a store waits for
      // the store's round trip -- the per-row load -> wait -> store chain costs one memory latency per 16 bytes
      // (igemm_patch.hip measured it).  Rows that store nothing load from pixel 0.
      int off[4];
      float4 xp[4], old[4];
      uchar4 ymk[4], gmk[4];
      const bool use_grad = p.grad_act != ADVOC_ACT_NONE;
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int pix = s_pix[di * BM + (wm * MT + i) * 32 + trow + 8 * ps];
        off[ps] = pix < 0 ? -1 : pix * d.c + ch;
        const int lo = pix < 0 ? ch : off[ps];
        if (p.y_mask) ymk[ps] = *reinterpret_cast<const uchar4*>(p.y_mask + lo);
        if (use_grad) xp[ps] = *reinterpret_cast<const float4*>(d.xpre + lo);
        if (d.gmask) gmk[ps] = *reinterpret_cast<const uchar4*>(d.gmask + lo);
        if (d.accum && !atomic_split) old[ps] = *reinterpret_cast<const float4*>(d.p + lo);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * half) * LDT + l32] = acc[i][j][r];
      wave_lds_sync();
      float4 gs = make_float4(1.f, 1.f, 1.f, 1.f), gh = make_float4(0.f, 0.f, 0.f, 0.f);
      if (use_grad && d.gscale) {
        gs = *reinterpret_cast<const float4*>(d.gscale + ch);
        gh = *reinterpret_cast<const float4*>(d.gshift + ch);
      }
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        if (off[ps] < 0) continue;
        float4 v = *reinterpret_cast<const float4*>(T + (trow + 8 * ps) * LDT + 4 * tq);
        v.x = fmaf(v.x, unscale, bias4.x); v.y = fmaf(v.y, unscale, bias4.y);
        v.z = fmaf(v.z, unscale, bias4.z); v.w = fmaf(v.w, unscale, bias4.w);
        if (p.y_mask) {
          const uchar4 mk = ymk[ps];
          v.x *= mk.x * p.y_mask_scale; v.y *= mk.y * p.y_mask_scale;
          v.z *= mk.z * p.y_mask_scale; v.w *= mk.w * p.y_mask_scale;
        }
        if (use_grad) {
          float4 x = xp[ps];
          x.x = x.x * gs.x + gh.x; x.y = x.y * gs.y + gh.y; x.z = x.z * gs.z + gh.z; x.w = x.w * gs.w + gh.w;
          v.x *= x.x > 0.f ? 1.f : gslope; v.y *= x.y > 0.f ? 1.f : gslope;
          v.z *= x.z > 0.f ? 1.f : gslope; v.w *= x.w > 0.f ? 1.f : gslope;
        }
        if (d.gmask) {
          const uchar4 mk = gmk[ps];
          v.x *= mk.x * d.gmask_scale; v.y *= mk.y * d.gmask_scale;
          v.z *= mk.z * d.gmask_scale; v.w *= mk.w * d.gmask_scale;
        }
        float* const dst = d.p + off[ps];
        if (atomic_split) {
          unsafeAtomicAdd(dst, v.x); unsafeAtomicAdd(dst + 1, v.y);
          unsafeAtomicAdd(dst + 2, v.z); unsafeAtomicAdd(dst + 3, v.w);
          continue;
        }
        if (d.accum) {
          v.x += old[ps].x; v.y += old[ps].y; v.z += old[ps].z; v.w += old[ps].w;
        }
        if (di == 1) dmax1 = fmaxf(fmaxf(dmax1, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        if (!no_store) *reinterpret_cast<float4*>(dst) = v;
        if (di == 0) {
          if (emit0) emit4(p.oimg[0], eup0, v, (unsigned)off[ps], evmax0);
          if (emit1) emit4(p.oimg[1], eup1, v, (unsigned)off[ps], evmax1);
          cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
        }
      }
      wave_lds_sync();
    }
    if (p.ocolsum_table && emit0 && di == 0) {
#pragma unroll
      for (int sh = 8; sh < 64; sh <<= 1) {
        cs.x += __shfl_xor(cs.x, sh, 64); cs.y += __shfl_xor(cs.y, sh, 64);
        cs.z += __shfl_xor(cs.z, sh, 64); cs.w += __shfl_xor(cs.w, sh, 64);
      }
      if (trow == 0) {
        float* row = p.ocolsum_table + (size_t)((blockIdx.x + blockIdx.z * 7) & (kColsumReplicas - 1)) * d.c + ch;
        unsafeAtomicAdd(row, cs.x); unsafeAtomicAdd(row + 1, cs.y); unsafeAtomicAdd(row + 2, cs.z); unsafeAtomicAdd(row + 3, cs.w);
      }
    }
  }
  if (emit0) emit_finish(p.oimg[0], eup0, evmax0);
  if (emit1) emit_finish(p.oimg[1], eup1, evmax1);
  if (p.d1_amax_out && !atomic_split) {      // max |value written to destination 1|: GatherGemmParams::d1_amax_out
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dmax1 = fmaxf(dmax1, __shfl_xor(dmax1, off, 64));
    if (lane == 0 && dmax1 > 0.f &&
        __float_as_uint(dmax1) > __hip_atomic_load(p.d1_amax_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(p.d1_amax_out, __float_as_uint(dmax1));
  }
}

template <int MT, int NT, int NS, int WGM>
__global__ __launch_bounds__(64 * WGM * 2, 2) void gather_gemm_h3_kernel(const GatherGemmParams p) {
  gather_gemm_h3_body<MT, NT, NS, WGM>(p);
}
// timing experiments only (ADVOC_H3_ABLATE=2..4): 2 no MFMAs (fragment reads kept), 3 no DMA, 4 DMA only
template <int MT, int NT, int NS, int ABL>
__global__ __launch_bounds__(256, 2) void gather_gemm_h3_abl_kernel(const GatherGemmParams p) {
  gather_gemm_h3_body<MT, NT, NS, 2, ABL>(p);
}

template <int MT, int NT, int NS, int WGM = 2>
int launch_h(const GatherGemmParams& p, hipStream_t stream, const char** name_only, const TailPlan& tail,
             float* tail_ws, int* tail_cnt, int ksplit) {
  using C = HCfg<MT, NT, NS, WGM>;
  if (name_only) {
    static const std::string name = std::string("gather_gemm_h3_kernel<") + std::to_string(MT) + ", " +
                                    std::to_string(NT) + ", " + std::to_string(NS) + ", " + std::to_string(WGM) + ">";
    *name_only = name.c_str();
    return ADVOC_OK;
  }
#ifdef ADVOC_DIAG
  static const int abl = getenv("ADVOC_H3_ABLATE") ? atoi(getenv("ADVOC_H3_ABLATE")) : 0;
#else
  constexpr int abl = 0;      // (timing experiments: -DADVOC_DIAG builds only, like ADVOC_H3_PATCH_ABLATE)
#endif
  auto kern = gather_gemm_h3_kernel<MT, NT, NS, WGM>;
  if constexpr (NS == 2 && MT == 2 && NT >= 2 && WGM == 2) {   // experiments: two instances are enough
    if (abl == 2) kern = gather_gemm_h3_abl_kernel<MT, NT, NS, 2>;
    if (abl == 3) kern = gather_gemm_h3_abl_kernel<MT, NT, NS, 3>;
    if (abl == 4) kern = gather_gemm_h3_abl_kernel<MT, NT, NS, 4>;
  }
  const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
  if (attr != hipSuccess) { note_hip_error(attr); return ADVOC_ERR_HIP; }
  const int64_t M = (int64_t)p.batch * p.gh * p.gw;
  const int64_t gx = ceil_div(M, C::BM) * ceil_div(p.n_total, C::BN);
  GatherGemmParams q = p;
  dim3 grid((unsigned)gx, (unsigned)ksplit, (unsigned)p.nphase);
#ifdef ADVOC_DIAG
  static const bool log_launches = getenv("ADVOC_H3_LOG") != nullptr;     // which launch is which (tools/micro)
#else
  constexpr bool log_launches = false;
#endif
  if (log_launches)
    fprintf(stderr, "h3 launch <%d,%d,%d,%d> batch %d grid %dx%d (+%d) phases %d N %d K %d taps %d bwd %d: %u x %u x %u wgs, "
            "tail main %d rem %d split %d\n", MT, NT, NS, WGM, p.batch, p.gh, p.gw, p.gx_off, p.nphase, p.n_total,
            p.c0 + p.c1, p.ntaps, p.grad_act != ADVOC_ACT_NONE || p.d[1].p != nullptr, grid.x, grid.y, grid.z, tail.main,
            tail.rem, tail.split);
  if (ksplit > 1) {
    // partial sums meet in the destination with fp32 atomics: start from zero (logical region only: columns
    // beyond out_w -- pitch padding -- stay untouched)
    for (int i = 0; i < 2; ++i) {
      const GemmDest& d = p.d[i];
      if (d.p == nullptr || d.accum) continue;
      // (a launch over grid columns [gx_off, gx_off + gw) only owns the output columns those map to)
      const int col0 = p.gx_off * p.osx;
      int cols = p.out_w - col0;
      if (p.gx_off && cols > p.gw * p.osx) cols = p.gw * p.osx;
      if (cols <= 0) continue;
      hipError_t e = hipMemset2DAsync(d.p + (size_t)col0 * d.c, sizeof(float) * (size_t)d.pitch * d.c, 0,
                                      sizeof(float) * (size_t)cols * d.c, (size_t)p.batch * p.out_h, stream);
      if (e != hipSuccess) { note_hip_error(e); return ADVOC_ERR_HIP; }
    }
  } else if (tail.split > 1 && tail_ws && tail_cnt) {
    q.tail_main = tail.main; q.tail_split = tail.split; q.tail_ws = tail_ws; q.tail_cnt = tail_cnt;
    grid = dim3((unsigned)(tail.main + tail.rem * tail.split), 1, 1);
  }
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(kern, grid, dim3(C::THREADS), C::LDS_BYTES, stream, q);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

struct Pick { int mt, nt, ns, wgm; };

// the 128 x 64 per-tap tile on 2, 3 or 4 LDS stages (ADVOC_H3_DEEP_STAGES)
int launch_h21(const GatherGemmParams& p, hipStream_t stream, const char** name_only, const TailPlan& tail,
               float* tail_ws, int* tail_cnt, int ksplit) {
  const int ns = tuning().h3_deep_stages;
  if (ns == 3) return launch_h<2, 1, 3>(p, stream, name_only, tail, tail_ws, tail_cnt, ksplit);
  if (ns == 4) return launch_h<2, 1, 4>(p, stream, name_only, tail, tail_ws, tail_cnt, ksplit);
  return launch_h<2, 1, 2>(p, stream, name_only, tail, tail_ws, tail_cnt, ksplit);
}

// Tile choice (tools/micro/h3_sweep.py, NOTEBOOK.md §4).  ADVOC_H3_TILE=1|2|3 forces 128x128 | 128x256 | 256x128,
// ADVOC_H3_STAGES=2|3 the LDS stages.
// ws_split: the caller's workspace can hold the parked K slices (the 256 x 256 / kept 128 x 128 choices for under-filled
// single-phase launches count on that split; without it they would run unsplit on a fraction of the chip).
Pick pick_tile(const GatherGemmParams& p, bool ws_split = true) {
  const Tuning& t = tuning();
  const int N = p.n_total;
  Pick k = {2, 2, 2, 2};
  if (N % 128 != 0 || t.h3_tile == 4) k = {2, 1, 2, 2};      // 64-column tiles (48 KiB: three workgroups per CU)
  else if (t.h3_tile == 5 && N % 256 == 0) k = {2, 4, 2, 4};  // 256 x 256, 8 waves
  // (r3: the 128 x 256 / 256 x 128 four-wave tiles, 256 x 128 on eight waves and the three-stage forms -- measured and
  // rejected in r2, NOTEBOOK.md section 7b -- are no longer compiled in: ADVOC_H3_TILE takes 1 | 4 | 5)
  else if (t.h3_tile == 0 && N % 256 == 0) {
    // 256 x 256 on 8 waves (one workgroup per CU) streams half the bytes per flop of 128 x 128 and measured 1.2-1.35x
    // faster on every launch with >= 2 such tiles per CU (381 vs 284 TFLOP/s on D layer_4); below that the launch would
    // fall into split-K, whose atomic epilogue on 64 Ki outputs per workgroup costs more than the tile saves
    // (re-measured without split-K, tools/micro/deep_sweep.py: from ONE such tile per CU on it is already 1.2 x the
    // 128 x 128 tile -- 422 vs 522 us on encoder_4 forward, 424 vs 526 on decoder_5 forward)
    const int64_t t256 = ceil_div((int64_t)p.batch * p.gh * p.gw, 256) * (N / 256) * p.nphase;
    const int cus = device_cu_count();
    if (!t.h3_deep_plan) {
      if (t256 >= cus) k = {2, 4, 2, 4};
    } else {
      // (r4, kernel-only times from a trace, profiles/r04_g_deep_split_*.txt) barely more than one such tile per CU is a
      // round and a nearly empty one: the four-phase launches of encoder_5 backward-data / decoder_5 forward (272 tiles on
      // 256 CUs) take 249 / 386 us on it and 220 / 385 us on 128 x 128 tiles; a single-phase launch with a quarter to half a
      // tile per CU and a long contraction (encoder_5 forward: 68 tiles x 256 K tiles) is cut into K slices that meet in
      // the workspace instead (launch_gather_gemm_h3): 224 us against 250 (128 x 128) and 291 (128 x 64)
      const int nkt = (p.c0 + p.c1) / 32 * p.ntaps;
      if (2 * t256 >= 3 * cus) k = {2, 4, 2, 4};
      else if (p.nphase == 1 && 4 * t256 >= cus && 2 * t256 <= cus && nkt >= 128 && t.igemm_splitk && ws_split) k = {2, 4, 2, 4};
    }
  }
  if (t.h3_tile == 0 && k.wgm == 2 && k.nt == 2) {
    // under one round of 128 x 128 tiles (two per CU) the 128 x 64 tile (three per CU) fills the chip better:
    // 1.04-1.2 x on the deep layers (encoder_5 / decoder_6 forward, encoder_6 / decoder_6 backward-data)
    // (r4: not for single-phase launches from half a round on -- decoder_6 backward-data, 144 tiles x 256 K tiles: 190 us on
    // 128 x 64, 129 us on 128 x 128 tiles cut into three K slices each)
    const int64_t t128 = ceil_div((int64_t)p.batch * p.gh * p.gw, 128) * ceil_div(N, 128) * p.nphase;
    const bool keep128 = t.h3_deep_plan && p.nphase == 1 && 2 * t128 >= device_cu_count() && t.igemm_splitk && ws_split;
    if (t128 < 2 * device_cu_count() && !keep128) k = {2, 1, 2, 2};
  }
  return k;
}

int weight_taps_of(const GatherGemmParams& p) {
  int taps = 0;
  for (int ph = 0; ph < p.nphase; ++ph)
    for (int i = 0; i < p.ntaps; ++i) taps = (p.tap[ph][i] >> 16) + 1 > taps ? (p.tap[ph][i] >> 16) + 1 : taps;
  return taps;
}

int64_t round256(int64_t b) { return (b + 255) / 256 * 256; }

}  // namespace

bool h3_eligible(const GatherGemmParams& p) {
  const Tuning& t = tuning();
  if (!t.h3 || !t.igemm_x6) return false;
  const int ktot = p.c0 + p.c1, N = p.n_total;
  // N = 32 (AdVoc-small's encoder_2 / layer_2 backward-data, decoder_2 forward: advoc_model_small.py:14-15) runs the
  // 128 x 64 tile with its upper 32 columns masked: half of the tile's MFMAs multiply zeros, which is still 3-4 x the
  // fp32 kernel these launches used to fall back to
  if (ktot % 32 || p.c0 % 32 || p.c1 % 32 || (N % 64 && N != 32) || p.n_split % 32) return false;
  if (p.n_valid && p.n_valid != p.n_total) return false;
  const int64_t rows128 = ceil_div((int64_t)p.batch * p.gh * p.gw, 128) * p.nphase;
  if (rows128 * ((N + 127) / 128) < t.h3_min_tiles) return false;
  // 32-bit byte offsets inside the kernel: either source image, and the weight image, below 2 GiB
  const int64_t lim = 0x7fffffffLL;
  const int64_t e0 = (int64_t)p.batch * p.a_h * p.a0_pitch * p.c0, e1 = (int64_t)p.batch * p.a_h * p.a1_pitch * p.c1;
  if (4 * e0 > lim || 4 * e1 > lim) return false;
  if ((int64_t)4 * weight_taps_of(p) * N * ktot > lim) return false;
  return true;
}

bool h3_weight_image_shape(const GatherGemmParams& p, int* taps, int* n_total, int* ktot) {
  if (!h3_eligible(p)) return false;
  *taps = weight_taps_of(p);
  *n_total = p.n_total;
  *ktot = p.c0 + p.c1;
  return true;
}

// Workspace layout of one launch: [operand headers 256 B][weight image][image of source 0][image of source 1]
// [tail partials]
int launch_gather_gemm_h3(const GatherGemmParams& p_in, bool b_kn, hipStream_t stream, const char** name_only,
                          float* scratch, int64_t scratch_bytes, int64_t* scratch_query) {
  if (!h3_eligible(p_in)) return ADVOC_ERR_UNSUPPORTED;
  GatherGemmParams p = p_in;
  const int ktot = p.c0 + p.c1, N = p.n_total;
  const int taps = weight_taps_of(p);
  const int64_t e0 = (int64_t)p.batch * p.a_h * p.a0_pitch * p.c0, e1 = (int64_t)p.batch * p.a_h * p.a1_pitch * p.c1;
  const int64_t hdr_bytes = 256 + kColsumBytes;       // operand headers + the bias-gradient replica table (image.hip)
  const int64_t wq_bytes = round256((int64_t)4 * taps * N * ktot);
  const int64_t i0_bytes = round256(4 * e0), i1_bytes = round256(4 * e1);
  PatchGeom geom;
  const int patch_nph = patch_plan(p, &geom);       // stride-1 gathers: igemm_patch.hip
  const int64_t need = hdr_bytes + wq_bytes + i0_bytes + i1_bytes;
  const int nkt = ktot / 32 * p.ntaps;
  Pick k;
  int BM, BN, ksplit;
  int64_t tiles, rtiles, tail_bytes;
  TailPlan tail;
  // planned twice at most: a workspace too small for the parked K slices (tail_bytes) re-plans without them -- tile choice
  // included -- instead of running the slices' tiles unsplit
  for (int ws_split = 1; ws_split >= 0; --ws_split) {
  k = pick_tile(p, ws_split != 0);
  BM = 32 * k.mt * k.wgm; BN = 64 * k.nt;
  tiles = ceil_div((int64_t)p.batch * p.gh * p.gw, BM) * ceil_div(N, BN) * p.nphase;
  // Small pixel grids with deep contractions (encoder_5.., decoder_5.. and their gradients) would leave CUs idle:
  // split K until the launch holds ~4 workgroups per CU, keeping >= 8 K tiles per slice (igemm.hip does the same)
  // (measured, tools/micro/deep_sweep.py / profiles/r02_deep_sweep.txt: with these kernels the atomic epilogue + the
  // zero-fill cost more than idle CUs down to a quarter of the chip -- encoder_6 backward-data 278 -> 116 us, decoder_6
  // forward 365 -> 213 us, encoder_5 forward 487 -> 393 us without the split; and never on 64 Ki-output tiles)
  ksplit = 1;
  tail = TailPlan();
  if (ws_split && !patch_nph && tiles < device_cu_count() && tiles <= 256 && (k.wgm == 2 || tuning().h3_deep_split > 0 || tuning().h3_deep_plan) &&
      tuning().igemm_splitk) {
    // Few tiles, deep contraction (the 8 x 17-point layers and below): every tile is cut into K slices that meet in the
    // WORKSPACE -- the tail-split mechanism with no whole tiles: the last slice to arrive sums the parked partial tiles
    // in slice order and runs the ordinary epilogue (no zero fill, no atomics, bias and activation gradient fused as
    // usual; <= 256 tiles: one arrival counter each)
    int split = (int)ceil_div((int64_t)tuning().h3_deep_wgs_per_cu * device_cu_count(), tiles);
    if (split > nkt / tuning().h3_deep_split_div) split = nkt / tuning().h3_deep_split_div;
    if (split > 16) split = 16;
    if (tuning().h3_deep_plan) {
      // (r4, profiles/r04_g_deep_split_*.txt) the rule above aims just past two workgroups per CU (576, 520, 640, 560 for the
      // model's layers: a round and a bit); the fastest split of every layer measured holds 430-480 workgroups of the 128-row
      // tiles, i.e. just UNDER two per CU, with at least 16 K tiles per slice (8 left the 64-K-tile four-phase launches of
      // encoder_8 / decoder_8 with more fixed cost than work); 256 x 256 tiles: one workgroup per CU
      const int cus = device_cu_count();
      split = k.wgm == 4 ? cus / (int)tiles : (int)((15 * cus / 8 + tiles / 2) / tiles);
      if (split > nkt / 16) split = nkt / 16;
      if (split > 16) split = 16;
    }
    if (tuning().h3_deep_split > 0) split = tuning().h3_deep_split < nkt / 2 ? tuning().h3_deep_split : nkt / 2;   // (experiments)
    if (split >= 2) { tail.main = 0; tail.rem = (int)tiles; tail.split = split; }
  }
  if (tail.split < 2 && !patch_nph && tiles < device_cu_count() / 2 && k.wgm == 2 && tuning().igemm_splitk) {
    // (without a workspace the slices meet in the destination with atomics, as in igemm.hip)
    ksplit = (int)ceil_div((int64_t)device_cu_count(), tiles);
    if (ksplit > nkt / 8) ksplit = nkt / 8;
    if (ksplit > 16) ksplit = 16;
    if (ksplit < 1) ksplit = 1;
  }
  if (ws_split && tail.split < 2 && ksplit == 1 && !patch_nph) tail = plan_tail(tiles, nkt);
  // the per-tap launch that takes the 1..4 remainder columns of a patch launch (below): one 128 x 64 tile per workgroup
  // and under one workgroup per CU it runs at the latency of a single K pipeline (~1 us per K tile, 60-90 us for 1-3 % of
  // the layer's work): its tiles are cut into K slices that meet in the workspace, like the deep layers' above
  rtiles = 0;
  if (patch_nph && geom.rem) {
    rtiles = ceil_div((int64_t)p.batch * p.gh * geom.rem, 128) * ceil_div(N, 64) * p.nphase;
    if (ws_split && tuning().h3_rem_ws && rtiles <= 256 && tuning().igemm_splitk) {
      int split = (int)ceil_div((int64_t)tuning().h3_rem_wgs_per_cu * device_cu_count(), rtiles);
      if (split > nkt / tuning().h3_rem_split_div) split = nkt / tuning().h3_rem_split_div;
      if (split > 16) split = 16;
      if (split >= 2) { tail.main = 0; tail.rem = (int)rtiles; tail.split = split; }
    }
  }
  tail_bytes = (int64_t)sizeof(float) * tail.rem * tail.split * (patch_nph ? 128 * 64 : BM * BN);
  if (scratch_query || !scratch || scratch_bytes >= need + tail_bytes || tail_bytes == 0) break;
  }
  if (scratch_query) { *scratch_query = need + tail_bytes; return ADVOC_OK; }
  if (!scratch || scratch_bytes < need) return ADVOC_ERR_UNSUPPORTED;
  // consumers' images from this launch's epilogue (image_emit.h): every epilogue must see the final value, so no K
  // split that meets in the destination with atomics (the workspace splits are fine: their last slice runs the epilogue)
  const bool want_emit = p.oimg[0].img != nullptr || p.oimg[1].img != nullptr;
  if (want_emit || p.d1_amax_out) ksplit = 1;      // (every epilogue must see the final value)
  if (p.emit_report) *p.emit_report = 1;
  // (r5) the output-gradient image of the layer below under the a-priori scale (GatherGemmParams::oimg_bounded): the patch
  // kernels' lean backward-data instances, grids without remainder columns
  // gates from the consuming layer's operand image (GemmDest::ximg): the patch kernels' backward-data instances, no
  // remainder columns
  if ((p.d[0].ximg || p.d[1].ximg) && (patch_nph == 0 || geom.rem != 0 || p.n_total == 32)) return ADVOC_ERR_UNSUPPORTED;
  const bool bounded_fwd = p.oimg_bounded && p.grad_act == ADVOC_ACT_NONE && !p.d[0].xpre && !p.d[1].p;
  if (p.oimg_bounded && !bounded_fwd) {
    // (the patch kernels' lean instances; remainder columns and launches without a patch plan on the per-tap kernel, whose
    // generic epilogue takes masks too -- but an accumulating destination 0 needs a bound of what it holds: obound_add)
    const bool ok = p.oimg[0].img && p.oimg[0].hdr && !p.oimg[1].img && !p.y_mask &&
                    !p.d[0].gmask && !p.d[1].gmask && (!p.d[0].accum || (p.obound_add && (patch_nph == 0 || patch_nph == 4 || patch_nph == 6))) &&
                    !p.d[1].accum &&
                    (p.grad_act != ADVOC_ACT_NONE || p.d[0].xpre || p.d[1].p) &&
                    (!p.ocolsum_out || (p.ocolsum_table && p.d[0].c <= 1024));
    if (p.emit_report) *p.emit_report = ok ? 3 : 0;
    if (!ok) return ADVOC_ERR_UNSUPPORTED;
  }
  if (bounded_fwd) {
    // a FORWARD launch that writes ONE consumer's image under the a-priori scale (fp32 output optional): patch kernels, grids
    // without remainder columns, no dropout mask on the output
    const bool ok = patch_nph != 0 && geom.rem == 0 && p.oimg[0].img && p.oimg[0].hdr && !p.oimg[1].img && !p.y_mask &&
                    !p.d[0].accum && p.n_total != 32;
    if (p.emit_report) *p.emit_report = ok ? 5 : 0;
    if (!ok) return ADVOC_ERR_UNSUPPORTED;
  }
  p.k_order = tuning().igemm_korder >= 0 ? tuning().igemm_korder : 1;
  char* ws = reinterpret_cast<char*>(scratch);
  unsigned* hdr_b = reinterpret_cast<unsigned*>(ws) + 2;
  unsigned* hdr_a = p.a_hdr_out ? p.a_hdr_out : reinterpret_cast<unsigned*>(ws);
  uint16_t* wq = reinterpret_cast<uint16_t*>(ws + hdr_bytes);
  // the A image: in the layer's persistent buffer when it has one (the weight-gradient call of the same step reads it
  // again, wgrad_h3.hip), else in the workspace; source 1 follows source 0 at its 256-byte-rounded size either way
  char* img_home = p.a_img_out ? reinterpret_cast<char*>(p.a_img_out) : ws + hdr_bytes + wq_bytes;
  uint16_t* img0 = reinterpret_cast<uint16_t*>(img_home);
  uint16_t* img1 = reinterpret_cast<uint16_t*>(img_home + i0_bytes);
  p.w_l1 = nullptr;
  if (p.w_img && p.w_img_hdr) {          // persistent, current weight image (advoc_weight_images_f32)
    wq = const_cast<uint16_t*>(p.w_img);
    hdr_b = const_cast<unsigned*>(p.w_img_hdr);
    if (p.w_img_l1) p.w_l1 = hdr_b;      // ... with the per-tap row-L1 maxima behind it (advoc_weight_images_l1_f32)
  }
  p.a_hdr = hdr_a; p.b_hdr = hdr_b;
  p.wq = wq;
  p.wq_taps = taps;
  p.a0_img = img0;
  p.a1_img = e1 ? img1 : nullptr;
  p.a0_img_bytes = (int)(4 * e0);
  p.a1_img_bytes = (int)(4 * e1);
  float* tail_ws = scratch_bytes >= need + tail_bytes ? reinterpret_cast<float*>(ws + need) : nullptr;
  int* tail_cnt = nullptr;
  if (!name_only && tuning().h3_skip_prep) {
    if (tail.split > 1 && tail_ws) tail_cnt = tail_counter_slot();
  } else if (!name_only) {
    int rc = ADVOC_OK;
    if (!(p.w_img && p.w_img_hdr)) {
      if (!p.w_amax) {
        hipError_t e = hipMemsetAsync(hdr_b, 0, 8, stream);
        if (e != hipSuccess) { note_hip_error(e); return ADVOC_ERR_HIP; }
      }
      rc = launch_pair_weights(p.w, wq, taps, N, ktot, b_kn, hdr_b, stream, p.w_amax);
      if (rc != ADVOC_OK) return rc;
    }
    if (p.a_img_out && p.a_img_current && p.a_img_emitted && !p.a_img_bounded) {
      // the producers' epilogues wrote this image under the one-pass scale: the refit check (exact re-image from the fp32
      // tensors when a value left the window) and the header rotation, as behind a one-pass image built here
      const ImageSource s0 = {p.a0, e0, p.c0, p.in_scale, p.in_shift, p.in_act, p.a_mask, p.a_mask_scale};
      const ImageSource s1 = {p.a1, e1, p.c1, p.in_scale ? p.in_scale + p.c0 : nullptr,
                              p.in_shift ? p.in_shift + p.c0 : nullptr, p.in_act, nullptr, 0.f};
      rc = launch_image_refit(s0, s1, img0, hdr_a, stream);
      if (rc != ADVOC_OK) return rc;
    }
    if (!(p.a_img_out && p.a_img_current)) {
      // one scale for the whole A operand: the largest magnitude over both sources of a channel concat
      const ImageSource s0 = {p.a0, e0, p.c0, p.in_scale, p.in_shift, p.in_act, p.a_mask, p.a_mask_scale};
      const ImageSource s1 = {p.a1, e1, p.c1, p.in_scale ? p.in_scale + p.c0 : nullptr,
                              p.in_shift ? p.in_shift + p.c0 : nullptr, p.in_act, nullptr, 0.f};
      rc = make_operand_image(s0, s1, img0, hdr_a, p.a_img_out && p.a_img_delayed, stream,
                              p.a_colsum && image_colsum_ok(p.c0) ? p.a_colsum : nullptr, p.in_w, p.a0_pitch,
                              reinterpret_cast<float*>(ws + 256), /*keep_history=*/p.a_hdr_out != nullptr);
      if (rc != ADVOC_OK) return rc;
    }
    if (tail.split > 1 && tail_ws) tail_cnt = tail_counter_slot();
  }
  // max |A operand| for the a-priori bound of a launch that writes the layer below's image: word 0 of a header the layer
  // above wrote under its own bound or of a per-call header, word 2 ("largest magnitude of the image in the buffer") of a
  // persistent header behind its rotation
  p.a_amax = hdr_a + ((p.a_img_bounded || !p.a_hdr_out) ? 0 : 2);
  if (patch_nph && bounded_fwd && !name_only) {
    hipError_t e = hipMemsetAsync(p.oimg[0].hdr, 0, 4, stream);      // the magnitude accumulator of the image
    if (e != hipSuccess) { note_hip_error(e); return ADVOC_ERR_HIP; }
    return launch_patch_gemm_h3(p, geom, patch_nph, stream, nullptr);
  }
  if (p.oimg_bounded && !name_only) {
    // backward-data under the bound: the magnitude word and the replica table are cleared once, every launch of the call (the
    // patches, their remainder columns -- or the per-tap launch alone) writes the image, the column sums are folded at the end
    hipError_t e = hipMemsetAsync(p.oimg[0].hdr, 0, 4, stream);
    if (e == hipSuccess && p.ocolsum_out)
      e = hipMemsetAsync(p.ocolsum_table, 0, sizeof(float) * kColsumReplicas * (size_t)p.d[0].c, stream);
    if (e != hipSuccess) { note_hip_error(e); return ADVOC_ERR_HIP; }
    int rc;
    if (patch_nph && geom.rem == 0) {
      rc = launch_patch_gemm_h3(p, geom, patch_nph, stream, nullptr);
    } else if (patch_nph) {
      GatherGemmParams pm = p;
      pm.gw = geom.px * 16;
      rc = launch_patch_gemm_h3(pm, geom, patch_nph, stream, nullptr);
      if (rc == ADVOC_OK) {
        GatherGemmParams pr = p;
        pr.gw = geom.rem;
        pr.gx_off = geom.px * 16;
        rc = (tail.split > 1 && tail_ws && tail_cnt) ? launch_h21(pr, stream, nullptr, tail, tail_ws, tail_cnt, 1)
                                                      : launch_h21(pr, stream, nullptr, TailPlan(), nullptr, nullptr, 1);
      }
    } else if (k.wgm == 4 && k.nt == 4) {
      rc = launch_h<2, 4, 2, 4>(p, stream, nullptr, tail, tail_ws, tail_cnt, 1);
    } else if (k.nt == 1) {
      rc = launch_h21(p, stream, nullptr, tail, tail_ws, tail_cnt, 1);
    } else {
      rc = launch_h<2, 2, 2>(p, stream, nullptr, tail, tail_ws, tail_cnt, 1);
    }
    if (rc == ADVOC_OK && p.ocolsum_out) rc = launch_colsum_reduce(p.ocolsum_table, p.ocolsum_out, p.d[0].c, stream);
    return rc;
  }
  if (patch_nph) {
    if (geom.rem == 0) return launch_patch_gemm_h3(p, geom, patch_nph, stream, name_only);
    // grid width = 16 px + a few columns (the model's 33 / 65 / 129 / 257-wide grids): the patches take the multiple of
    // 16, a per-tap launch the remaining columns -- a whole patch column for 1 of 16 columns costs 1.1-1.5 x the work
    GatherGemmParams pm = p;
    pm.gw = geom.px * 16;
    const int rc = launch_patch_gemm_h3(pm, geom, patch_nph, stream, name_only);
    if (rc != ADVOC_OK || name_only) return rc;
    GatherGemmParams pr = p;
    pr.gw = geom.rem;
    pr.gx_off = geom.px * 16;
    // few rows, the whole contraction: split K over workgroups (zero fill + atomics) only when the contraction is long
    // and the launch would leave most of the chip idle
    int rsplit = 1;
    const Tuning& tn = tuning();
    if (tail.split > 1 && tail_ws && tail_cnt)
      return launch_h21(pr, stream, nullptr, tail, tail_ws, tail_cnt, 1);
    if (!want_emit && !p.d1_amax_out && nkt >= 4 * tn.h3_rem_split_div && rtiles < (int64_t)tn.h3_rem_wgs_per_cu * device_cu_count() && tn.igemm_splitk) {
      rsplit = (int)ceil_div((int64_t)tn.h3_rem_wgs_per_cu * device_cu_count(), rtiles);
      if (rsplit > nkt / tn.h3_rem_split_div) rsplit = nkt / tn.h3_rem_split_div;
      if (rsplit > 16) rsplit = 16;
      if (rsplit < 1) rsplit = 1;
    }
    return launch_h21(pr, stream, nullptr, TailPlan(), nullptr, nullptr, rsplit);
  }
  if (k.wgm == 4 && k.nt == 4) return launch_h<2, 4, 2, 4>(p, stream, name_only, tail, tail_ws, tail_cnt, ksplit);
  if (k.nt == 1) return launch_h21(p, stream, name_only, tail, tail_ws, tail_cnt, ksplit);
  return launch_h<2, 2, 2>(p, stream, name_only, tail, tail_ws, tail_cnt, ksplit);
}

}  // namespace advoc
