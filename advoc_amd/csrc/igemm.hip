// Gather-GEMM (implicit GEMM) convolution kernel for gfx950, exact fp32 on the matrix cores.
//
// Replaces the cuDNN / Eigen Conv2D, Conv2DBackpropInput kernels TF1 runs for
// models/advoc/advoc_model.py:25-69 (layers) in both directions.  See igemm.h for the GEMM view.
//
// Mapping to CDNA4
//   * v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate): bit-identical to an fmaf chain, the only
//     matrix path that meets the 1e-4 relative-L2 parity bar through 16 stacked convs.
//   * workgroup = 4 wavefronts (one per SIMD), block tile BM x BN = (32 MT WGM) x (32 NT WGN),
//     K step 16.  Each wave owns MT x NT accumulators of 32x32 (16 VGPRs each).
//   * A (im2col rows) is never materialised: each thread gathers float4s along the channel
//     axis straight from the NHWC activation(s), applies the fused prologue (BN affine, leaky /
//     plain ReLU, dropout mask of the backward pass) in registers and parks the tile in LDS as
//     As[m][k] (row stride 20 floats: conflict-free ds_write_b128 / ds_read_b128).  The skip
//     concat and the [:, :, :-1, :] trims are pointer / bound arithmetic.
//   * the two 32-lane halves of a wave consume K slots {0..7} and {8..15} of the 16-deep tile
//     (A and B use the same assignment), so one ds_read_b128 feeds four MFMAs.
//   * global loads for tile t+1 are issued before the 8*MT*NT MFMAs of tile t and written to the
//     other LDS buffer after them: one s_barrier per K tile, HBM/L2 latency hidden under MFMA.
//   * epilogue: bias, dropout mask, act'(x) gating, optional accumulate, split over two
//     destinations (skip-connection gradients); 128 B contiguous per 32-lane store.
#include <stdlib.h>

#include <atomic>
#include <string>

#include "common.h"
#include "igemm.h"
#include "tuning.h"
#include "x6.h"

namespace advoc {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// K tile depth BK (16 or 32) is a template parameter; LDS rows are BK + 4 floats (conflict-free
// ds_write_b128 / ds_read_b128 for both depths).

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == ADVOC_ACT_LRELU02) return fmaxf(0.2f * v, v);
  if (act == ADVOC_ACT_RELU) return fmaxf(v, 0.f);
  return v;
}

__device__ __forceinline__ float act_grad(float x, int act) {
  // derivative of tf.maximum(0.2x, x) / tf.nn.relu as TF defines them: MaximumGrad routes a tie
  // (x == 0) to the first argument (0.2x); ReluGrad passes gradient only where x > 0.
  if (act == ADVOC_ACT_LRELU02) return x > 0.f ? 1.f : 0.2f;
  if (act == ADVOC_ACT_RELU) return x > 0.f ? 1.f : 0.f;
  return 1.f;
}

template <int MT, int NT, int WGM, int WGN, bool B_KN, int BK, bool X6 = false>
struct Cfg {
  static constexpr int BM = 32 * MT * WGM;
  static constexpr int BN = 32 * NT * WGN;
  static constexpr int LDA = BK + 4;
  static constexpr int QPR = BK / 4;                       // float4 per tile row
  static constexpr int LDB_KN = BN + 4;
  static constexpr int A_TILE = BM * LDA;
  static constexpr int B_TILE = B_KN ? BK * LDB_KN : BN * LDA;
  static constexpr int A_LOADS = BM * QPR / 256;           // float4 per thread per K tile
  static constexpr int B_LOADS = (BN * QPR + 255) / 256;   // float4 per thread per K tile
  // Exactly the two double-buffered tiles: 40 KiB for 128x128 so FOUR workgroups fit a CU's
  // 160 KiB.  The epilogue's pixel table / transpose patches reuse the tile buffers once the K loop
  // is done.
  // split-bf16 path: three bf16 planes per operand, rows of 16 contraction slots = 8 dwords
  static constexpr int XA = 3 * BM * 8;                    // dwords per A stage
  static constexpr int XB = 3 * BN * 8;
  static constexpr size_t LDS_BYTES = X6 ? sizeof(float) * 2 * (XA + XB) : sizeof(float) * (2 * A_TILE + 2 * B_TILE);
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// act(v) = max(v, slope * v): slope 1 -> identity, 0.2 -> leaky ReLU, 0 -> ReLU (branch-free)
__device__ __forceinline__ float act_slope(int act) {
  return act == ADVOC_ACT_LRELU02 ? 0.2f : (act == ADVOC_ACT_RELU ? 0.f : 1.f);
}

// launch_bounds(256, 2): budget registers for 2 waves per SIMD (<= 256 VGPR+AGPR).  With the
// default bound hipcc chases a higher occupancy and spills the prefetch registers to scratch,
// which serialises the global loads behind s_waitcnt vmcnt(0).
template <int MT, int NT, int WGM, int WGN, bool B_KN, int BK, bool X6 = false>
__global__ __launch_bounds__(256, 2) void gather_gemm_kernel(const GatherGemmParams p) {
  using C = Cfg<MT, NT, WGM, WGN, B_KN, BK, X6>;
  static_assert(!X6 || (BK == 16 && !B_KN), "split-bf16 path: K tile of 16, pre-split weights");
  constexpr int BM = C::BM, BN = C::BN, LDA = C::LDA, QPR = C::QPR;
  constexpr int AL = C::A_LOADS, BL = C::B_LOADS;
  constexpr int RPP = 256 / QPR;   // tile rows covered by one pass of the 256 loader threads
  static_assert(WGM * WGN == 4, "4 wavefronts per workgroup");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                               // [2][BM][LDA]
  float* Bs = smem + 2 * C::A_TILE;               // [2][B_TILE]
  // epilogue tables, carved from the tile buffers AFTER the last K-loop barrier:
  //   [0, 4*32*36) floats  per-wave transpose patches;  then [2][BM] ints pixel table
  int* s_pix = reinterpret_cast<int*>(smem + 4 * 32 * 36);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  // grid points: < 2^31 (checked by the launcher), so the (row -> image, y, x) splits below are
  // 32-bit divisions; as int64 they were ~a third of this kernel's VALU work on shallow layers
  const unsigned M = (unsigned)p.batch * (unsigned)p.gh * (unsigned)p.gw;
  // XCD-aware tile order.  Workgroup b runs on XCD b % 8 (each XCD has its own L2); remap so that
  // an XCD walks a CONTIGUOUS run of tiles: neighbouring M tiles share input halo rows and the
  // N tiles of one M tile share the whole A tile, so those re-reads hit the local L2.
  const int ntn = p.n_total / BN;                       // N tiles
  // Tail split (p.tail_main > 0, 1-D launch over all phases): the first tail_main tiles -- whole
  // rounds of the chip -- run as usual; each of the remaining tiles is cut into tail_split K
  // slices, one workgroup each, in launch order AFTER the whole tiles and NOT remapped, so the
  // slices spread over every XCD / CU instead of giving a few CUs one more whole tile.
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
  const int kpt = ktot / BK;         // K tiles per tap
  const int nkt = kpt * p.ntaps;
  const float slope = act_slope(p.in_act);

  // ---- per-thread A rows (fixed for the whole K loop); 32-bit element offsets ----
  const int kq = tid % QPR;  // which float4 of the BK-wide K slice
  int row_y[AL], row_x[AL], base0[AL], base1[AL];
  bool row_ok[AL];
#pragma unroll
  for (int i = 0; i < AL; ++i) {
    const unsigned m = m0 + tid / QPR + RPP * i;
    row_ok[i] = m < M;
    const unsigned mm = row_ok[i] ? m : 0u;
    const unsigned t = mm / (unsigned)p.gw;
    const int gx = (int)(mm - t * (unsigned)p.gw);
    const int img = (int)(t / (unsigned)p.gh);
    const int gy = (int)(t - (unsigned)img * (unsigned)p.gh);
    row_x[i] = gx * p.sx;
    row_y[i] = gy * p.sy;
    base0[i] = ((img * p.a_h + row_y[i]) * p.a0_pitch + row_x[i]) * p.c0 + 4 * kq;
    base1[i] = ((img * p.a_h + row_y[i]) * p.a1_pitch + row_x[i]) * p.c1 + 4 * kq - base0[i];
  }   // base1 holds the DIFFERENCE to base0: offset = base0 + sel * base1 (no array select -> no scratch)
  // ---- per-thread B slots ----
  // B_KN: slot = (k row, 4 consecutive n); else slot = (n row, 4 consecutive k).  Threads beyond
  // the tile (BN = 32) re-read slot 0..: harmless, they skip the LDS store.
  int b_goff[BL], b_loff[BL];
  bool b_store[BL], b_live[BL];
  const int n_valid = p.n_valid ? p.n_valid : p.n_total;
#pragma unroll
  for (int i = 0; i < BL; ++i) {
    const int idx = tid + 256 * i;
    b_store[i] = idx < BN * QPR;
    const int id = b_store[i] ? idx : idx % (BN * QPR);
    if (B_KN) {
      const int k = id / (BN / 4), nq = id % (BN / 4);
      b_goff[i] = k * p.n_total + n0 + 4 * nq;           // + (wtap*ktot + k0) * N per tile
      b_loff[i] = k * C::LDB_KN + 4 * nq;
      b_live[i] = n0 + 4 * nq < n_valid;
    } else {
      const int n = id / QPR, q = id % QPR;
      b_goff[i] = (n0 + n) * ktot + 4 * q;                // + wtap*N*ktot + k0 per tile
      b_loff[i] = n * LDA + 4 * q;
      b_live[i] = n0 + n < n_valid;
    }
  }

  // split-bf16 path: this thread's 16 bytes of the pre-split weight tile (row n0 + tid / 2, half tid & 1)
  const int wq_row = (n0 + (tid >> 1)) * ktot + 8 * (tid & 1);
  float4 ra[AL], rb[BL];
  uint4 rbx0 = make_uint4(0, 0, 0, 0), rbx1 = rbx0, rbx2 = rbx0;   // split-bf16 path: 16 bytes per weight plane
  uint4 rby0 = rbx0, rby1 = rbx0, rby2 = rbx0;                     // ... and of rows 128.. of a 256-column tile
  unsigned a_ok = 0;
  int a_chan = 0;
  bool a_first = true;     // the staged A slice comes from source 0 (the only one a_mask covers)
  int a_moff[AL];

  // K order: the loader walks (tap, channel slice) pairs incrementally.  taps_inner: all taps of
  // one BK-channel slice before the next slice -- a workgroup then re-touches the same input
  // pixels on consecutive K tiles (L2 hits) instead of once per pass over the channels.
  // split-K: blockIdx.y owns K tiles [kt_begin, kt_end); partial results are combined with
  // hardware fp32 atomics in the epilogue (launch_cfg zeroes the destination first)
  const int kt_begin = (int)((int64_t)nkt * ks_idx / ks_cnt);
  const int kt_end = (int)((int64_t)nkt * (ks_idx + 1) / ks_cnt);
  const bool taps_inner = p.k_order != 0;
  int ld_tap = taps_inner ? kt_begin % p.ntaps : kt_begin / kpt;
  int ld_k0 = (taps_inner ? kt_begin / p.ntaps : kt_begin % kpt) * BK;
#define ADVOC_LOAD_TILE(KT)                                                                          \
  {                                                                                                  \
    const int ti_ = ld_tap;                                                                          \
    const int k0_ = ld_k0;                                                                           \
    if (taps_inner) {                                                                                \
      if (++ld_tap == p.ntaps) { ld_tap = 0; ld_k0 += BK; if (ld_k0 == ktot) ld_k0 = 0; }            \
    } else {                                                                                         \
      ld_k0 += BK;                                                                                   \
      if (ld_k0 == ktot) { ld_k0 = 0; if (++ld_tap == p.ntaps) ld_tap = 0; }                         \
    }                                                                                                \
    /* tap word straight from the kernel arguments: a scalar load, so the tap / weight-slab offsets */ \
    /* below are SALU work (through LDS they were 64-bit VALU multiplies on every K tile)          */ \
    const int tp_ = __builtin_amdgcn_readfirstlane(p.tap[phase][ti_]);                               \
    const int dy_ = (int)(int8_t)(tp_ & 0xff), dx_ = (int)(int8_t)((tp_ >> 8) & 0xff);              \
    const int wtap_ = tp_ >> 16;                                                                     \
    const bool second_ = k0_ >= p.c0;                                                                \
    const int sel_ = second_ ? 1 : 0;                                                                \
    const float* src_ = second_ ? p.a1 : p.a0;                                                       \
    const int delta_ = second_ ? (dy_ * p.a1_pitch + dx_) * p.c1 + (k0_ - p.c0)                      \
                               : (dy_ * p.a0_pitch + dx_) * p.c0 + k0_;                              \
    a_chan = k0_ + 4 * kq;                                                                           \
    a_first = !second_;                                                                              \
    a_ok = 0;                                                                                        \
    _Pragma("unroll") for (int i = 0; i < AL; ++i) {                                                 \
      const int iy_ = row_y[i] + dy_, ix_ = row_x[i] + dx_;                                          \
      const bool ok_ = row_ok[i] && (unsigned)iy_ < (unsigned)p.in_h && (unsigned)ix_ < (unsigned)p.in_w; \
      const int off_ = base0[i] + sel_ * base1[i] + delta_;                                          \
      a_moff[i] = off_;                                                                              \
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);                                                       \
      if (ok_) {                                                                                     \
        ra[i] = *reinterpret_cast<const float4*>(src_ + off_);                                       \
        a_ok |= 1u << i;                                                                             \
      }                                                                                              \
    }                                                                                                \
    if constexpr (X6) {                                                                              \
      /* pre-split weights: 16 bytes = 8 contraction slots of one output channel, per plane */      \
      const uint16_t* wq_ = p.wq + ((int64_t)wtap_ * p.n_total * ktot + k0_) + wq_row;               \
      const int64_t plane_ = (int64_t)p.wq_taps * p.n_total * ktot;                                  \
      if (tid < 2 * BN) {                                                                            \
        rbx0 = *reinterpret_cast<const uint4*>(wq_);                                                 \
        rbx1 = *reinterpret_cast<const uint4*>(wq_ + plane_);                                        \
        rbx2 = *reinterpret_cast<const uint4*>(wq_ + 2 * plane_);                                    \
      }                                                                                              \
      if constexpr (BN > 128) {   /* second half of a 256-column tile: rows n0 + 128 + tid / 2 */    \
        rby0 = *reinterpret_cast<const uint4*>(wq_ + 128 * ktot);                                    \
        rby1 = *reinterpret_cast<const uint4*>(wq_ + 128 * ktot + plane_);                           \
        rby2 = *reinterpret_cast<const uint4*>(wq_ + 128 * ktot + 2 * plane_);                       \
      }                                                                                              \
    } else {                                                                                         \
    const float* wb_ = B_KN ? p.w + ((int64_t)wtap_ * ktot + k0_) * p.n_total                        \
                            : p.w + (int64_t)wtap_ * n_valid * ktot + k0_;                           \
    _Pragma("unroll") for (int i = 0; i < BL; ++i) {                                                 \
      rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);                                                       \
      if (b_live[i]) rb[i] = *reinterpret_cast<const float4*>(wb_ + b_goff[i]);                      \
    }                                                                                                \
    }                                                                                                \
  }

#define ADVOC_STORE_TILE(BUF)                                                                        \
  {                                                                                                  \
    float* Ab_ = As + (BUF) * C::A_TILE;                                                             \
    float* Bb_ = Bs + (BUF) * C::B_TILE;                                                             \
    float4 sc_ = make_float4(1.f, 1.f, 1.f, 1.f), sh_ = make_float4(0.f, 0.f, 0.f, 0.f);            \
    if (p.in_scale) {                                                                                \
      sc_ = *reinterpret_cast<const float4*>(p.in_scale + a_chan);                                   \
      sh_ = *reinterpret_cast<const float4*>(p.in_shift + a_chan);                                   \
    }                                                                                                \
    _Pragma("unroll") for (int i = 0; i < AL; ++i) {                                                 \
      float4 v = ra[i];                                                                              \
      const float live_ = ((a_ok >> i) & 1u) ? 1.f : 0.f;   /* zero padding stays zero */            \
      if (p.in_scale) {   /* uniform: only batch-norm models carry an affine */                      \
      v.x = fmaf(v.x, sc_.x, sh_.x * live_); v.y = fmaf(v.y, sc_.y, sh_.y * live_);                  \
      v.z = fmaf(v.z, sc_.z, sh_.z * live_); v.w = fmaf(v.w, sc_.w, sh_.w * live_);                  \
      }                                                                                              \
      if (slope != 1.f) {   /* uniform: backward-data gathers a gradient, no activation */          \
      v.x = fmaxf(v.x, slope * v.x); v.y = fmaxf(v.y, slope * v.y);                                  \
      v.z = fmaxf(v.z, slope * v.z); v.w = fmaxf(v.w, slope * v.w);                                  \
      }                                                                                              \
      if (p.a_mask && a_first) {                                                                     \
        uchar4 mk_ = make_uchar4(0, 0, 0, 0);                                                        \
        if ((a_ok >> i) & 1u) mk_ = *reinterpret_cast<const uchar4*>(p.a_mask + a_moff[i]);          \
        v.x *= mk_.x * p.a_mask_scale; v.y *= mk_.y * p.a_mask_scale;                                \
        v.z *= mk_.z * p.a_mask_scale; v.w *= mk_.w * p.a_mask_scale;                                \
      }                                                                                              \
      if constexpr (X6) {                                                                            \
        unsigned* Ax_ = reinterpret_cast<unsigned*>(smem) + (BUF) * (C::XA + C::XB);                 \
        unsigned h0_[4], h1_[4], h2_[4];                                                             \
        split3(v.x, h0_[0], h1_[0], h2_[0]); split3(v.y, h0_[1], h1_[1], h2_[1]);                    \
        split3(v.z, h0_[2], h1_[2], h2_[2]); split3(v.w, h0_[3], h1_[3], h2_[3]);                    \
        const int row_ = tid / QPR + RPP * i;                                                        \
        *reinterpret_cast<uint2*>(Ax_ + (0 * BM + row_) * 8 + 2 * kq) =                              \
            make_uint2(pack_hi16(h0_[0], h0_[1]), pack_hi16(h0_[2], h0_[3]));                        \
        *reinterpret_cast<uint2*>(Ax_ + (1 * BM + row_) * 8 + 2 * kq) =                              \
            make_uint2(pack_hi16(h1_[0], h1_[1]), pack_hi16(h1_[2], h1_[3]));                        \
        *reinterpret_cast<uint2*>(Ax_ + (2 * BM + row_) * 8 + 2 * kq) =                              \
            make_uint2(pack_hi16(h2_[0], h2_[1]), pack_hi16(h2_[2], h2_[3]));                        \
      } else {                                                                                       \
      *reinterpret_cast<float4*>(Ab_ + (tid / QPR + RPP * i) * LDA + 4 * kq) = v;                    \
      }                                                                                              \
    }                                                                                                \
    if constexpr (X6) {                                                                              \
      unsigned* Bx_ = reinterpret_cast<unsigned*>(smem) + (BUF) * (C::XA + C::XB) + C::XA;           \
      if (tid < 2 * BN) {                                                                            \
        *reinterpret_cast<uint4*>(Bx_ + 4 * tid) = rbx0;                                             \
        *reinterpret_cast<uint4*>(Bx_ + BN * 8 + 4 * tid) = rbx1;                                    \
        *reinterpret_cast<uint4*>(Bx_ + 2 * BN * 8 + 4 * tid) = rbx2;                                \
      }                                                                                              \
      if constexpr (BN > 128) {                                                                      \
        *reinterpret_cast<uint4*>(Bx_ + 1024 + 4 * tid) = rby0;                                      \
        *reinterpret_cast<uint4*>(Bx_ + BN * 8 + 1024 + 4 * tid) = rby1;                             \
        *reinterpret_cast<uint4*>(Bx_ + 2 * BN * 8 + 1024 + 4 * tid) = rby2;                         \
      }                                                                                              \
    } else {                                                                                         \
    _Pragma("unroll") for (int i = 0; i < BL; ++i)                                                   \
        if (b_store[i]) *reinterpret_cast<float4*>(Bb_ + b_loff[i]) = rb[i];                         \
    }                                                                                                \
  }

  floatx16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int half = lane >> 5, l32 = lane & 31;

  ADVOC_LOAD_TILE(kt_begin);
  ADVOC_STORE_TILE(0);
  __syncthreads();

  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int buf = (kt - kt_begin) & 1;
    const bool more = kt + 1 < kt_end;
    // unconditional prefetch (the last iteration wraps round to tile 0 and drops it): keeps the
    // prefetch registers out of a conditional region
    ADVOC_LOAD_TILE(kt + 1);

    if constexpr (X6) {
      // six bf16 products per 32x32x16 block, smallest terms first: a1 b1, a0 b2, a2 b0, a0 b1, a1 b0, a0 b0
      // (the dropped a1 b2, a2 b1, a2 b2 are <= 2^-22 of the product).  Lane (l32, half) holds
      // contraction slots [8 half, 8 half + 8) of row / column l32.
      const unsigned* Ax = reinterpret_cast<const unsigned*>(smem) + buf * (C::XA + C::XB);
      const unsigned* Bx = Ax + C::XA;
      bf16x8 af[MT][3], bq[NT][3];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          af[i][pl] = *reinterpret_cast<const bf16x8*>(Ax + (pl * BM + (wm * MT + i) * 32 + l32) * 8 + 4 * half);
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          bq[j][pl] = *reinterpret_cast<const bf16x8*>(Bx + (pl * BN + (wn * NT + j) * 32 + l32) * 8 + 4 * half);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bq[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bq[j][2], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bq[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bq[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bq[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bq[j][0], acc[i][j], 0, 0, 0);
        }
    } else {
    const float* Ab = As + buf * C::A_TILE;
    const float* Bb = Bs + buf * C::B_TILE;
    constexpr int KH = BK / 2;   // K slots per wave half: lanes 0-31 take [0, KH), lanes 32-63 [KH, BK)
    float a[MT][KH], b[NT][KH];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const float* q = Ab + ((wm * MT + i) * 32 + l32) * LDA + half * KH;
#pragma unroll
      for (int c = 0; c < KH / 4; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(q + 4 * c);
        a[i][4 * c] = v.x; a[i][4 * c + 1] = v.y; a[i][4 * c + 2] = v.z; a[i][4 * c + 3] = v.w;
      }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = (wn * NT + j) * 32 + l32;
      if (B_KN) {
#pragma unroll
        for (int s = 0; s < KH; ++s) b[j][s] = Bb[(half * KH + s) * C::LDB_KN + col];
      } else {
        const float* q = Bb + col * LDA + half * KH;
#pragma unroll
        for (int c = 0; c < KH / 4; ++c) {
          const float4 v = *reinterpret_cast<const float4*>(q + 4 * c);
          b[j][4 * c] = v.x; b[j][4 * c + 1] = v.y; b[j][4 * c + 2] = v.z; b[j][4 * c + 3] = v.w;
        }
      }
    }
#pragma unroll
    for (int s = 0; s < KH; ++s)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
    }

    if (more) ADVOC_STORE_TILE(buf ^ 1);
    __syncthreads();
  }
#undef ADVOC_LOAD_TILE
#undef ADVOC_STORE_TILE

  // ---- tail slices: park the partial tile; the LAST slice to arrive sums all of them in slice
  // order (run-to-run reproducible) and carries on into the ordinary epilogue.  The partials and
  // the counter move with device-coherent (agent-scope atomic, sc1) accesses, which go past the
  // per-XCD L2s: a __threadfence() pair would do too, but it writes back and invalidates a whole
  // L2 per wave, which on ~1000 waves cost more than the balance gained (measured). ----
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

  // ---- epilogue ----
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

  const float gslope = act_slope(p.grad_act);   // act'(x) = x > 0 ? 1 : slope
  // Each 32x32 accumulator tile is transposed through a private LDS patch so that global
  // traffic is 16 bytes per lane on full 128-byte channel rows (8 lanes per pixel): the MFMA C
  // layout itself would give 4-byte stores, ~3x slower on the store-heavy narrow layers.
  constexpr int LDT = 36;
  float* T = smem + wave * (32 * LDT);          // the K-loop buffers are dead after the last barrier
  const int trow = lane >> 3, tq = lane & 7;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int nt0 = n0 + (wn * NT + j) * 32;    // first channel of this 32-wide tile
    const int di = nt0 >= p.n_split ? 1 : 0;    // uniform: n_split % 32 == 0
    const GemmDest& d = p.d[di];
    if (d.p == nullptr || nt0 >= n_valid) continue;
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
        if (pix < 0 || nt0 + 4 * tq >= n_valid) continue;
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
        if (atomic_split) {   // every epilogue factor above is linear in the accumulator
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

// ADVOC_IGEMM_SPLITK=0 keeps every launch on one K pass (bitwise run-to-run reproducible results;
// split-K combines partial sums with atomics, whose order varies).  Read per launch so a test
// can toggle it.
bool split_k_allowed() { return tuning().igemm_splitk != 0; }

struct LaunchCtx {
  hipStream_t stream;
  const char** name_only;    // report the kernel instance instead of launching
  float* scratch;            // caller workspace for the tail split (may be null)
  int64_t scratch_bytes;
  int64_t* scratch_query;    // report the workspace bytes wanted instead of launching
};

// ADVOC_IGEMM_TAIL=0 disables the tail split (A/B measurements).
bool tail_split_allowed() { return tuning().igemm_tail != 0; }

}  // namespace

int device_cu_count() {
  static const int cus = [] {
    int dev = 0, n = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    (void)hipGetLastError();
    return n > 0 ? n : 256;
  }();
  return cus;
}

// CUs a persistent launch (one workgroup per CU walking tiles) may fill: all of them, minus ADVOC_RESERVE_CUS (data-parallel
// runs leave a few for RCCL's kernels), in whole rows of the 8 XCDs
int persistent_cu_count() {
  int n = device_cu_count() - tuning().reserve_cus;
  n = n / 8 * 8;
  return n >= 8 ? n : 8;
}

// Arrival counters of the tail split: a ring of slots so that launches in flight on different
// streams do not share counters; every launch leaves its slot zeroed again (see the kernel).
constexpr int kTailSlots = 64;
constexpr int kTailMaxTiles = 256;
__device__ int g_tail_cnt[kTailSlots * kTailMaxTiles];

int* tail_counter_slot() {
  static std::atomic<unsigned> next{0};
  int* base = nullptr;
  if (hipGetSymbolAddress(reinterpret_cast<void**>(&base), HIP_SYMBOL(g_tail_cnt)) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return base + (next.fetch_add(1) % kTailSlots) * kTailMaxTiles;
}

// A launch of T equal workgroups on C compute units costs ceil(T / C) rounds when they are all
// resident (measured: 1024 tiles 193 us, 1056 tiles 230 us on 256 CUs).  Plan: the last T mod C
// tiles are cut into `split` K slices each so that the extra round is 1/split of a tile long.
TailPlan plan_tail(int64_t tiles, int nkt) {
  TailPlan t;
  const int cus = device_cu_count();
  if (!tail_split_allowed() || tiles < cus || tiles > 0x3fffffffLL) return t;
  const int rem = (int)(tiles % cus);
  if (rem == 0 || rem > kTailMaxTiles) return t;
  int split = cus / rem;
  if (split > 8) split = 8;
  while (split > 1 && nkt / split < 8) --split;
  if (split < 2) return t;
  t.main = (int)(tiles - rem); t.rem = rem; t.split = split;
  return t;
}

namespace {

int weight_taps(const GatherGemmParams& p) {
  int t = 0;
  for (int ph = 0; ph < p.nphase; ++ph)
    for (int i = 0; i < p.ntaps; ++i) t = (p.tap[ph][i] >> 16) + 1 > t ? (p.tap[ph][i] >> 16) + 1 : t;
  return t;
}

// Large launches run on the bf16 matrix path with every operand split into three bf16 terms (fp32-level
// error, see split3); needs caller workspace for the split weights.  ADVOC_IGEMM_X6=0 keeps everything
// on the fp32 MFMA kernels (A/B measurements).
bool x6_allowed() { return tuning().igemm_x6 != 0; }

// 128 x 256 split tile (A is split once per 256 output columns; 256 registers, two workgroups per CU):
// measured 6-20 % faster than 128 x 128 on deep contractions with >= ~500 such tiles (D layer_4, the
// 512-channel generator layers of the full model), 10-150 % slower on shallow ones or small grids.
// ADVOC_IGEMM_X6_WIDE=<tiles> overrides the tile threshold and drops the depth condition (experiments).
bool x6_wide(int64_t tiles256, int k_total, int nphase) {
  if (tuning().igemm_x6_wide >= 0) return tiles256 >= tuning().igemm_x6_wide;
  // ... and on single-phase launches (forward / stride-1 layers) already from 2 048 deep with >= 900 tiles
  return (tiles256 >= 500 && k_total >= 4096) || (nphase == 1 && tiles256 >= 900 && k_total >= 2048);
}

template <int MT, int NT, int WGM, int WGN, bool B_KN, int BK, bool X6 = false>
int launch_cfg(const GatherGemmParams& p, const LaunchCtx& ctx, bool b_kn_src = B_KN) {
  using C = Cfg<MT, NT, WGM, WGN, B_KN, BK, X6>;
  hipStream_t stream = ctx.stream;
  const char** name_only = ctx.name_only;
  float* scratch = ctx.scratch;
  int64_t scratch_bytes = ctx.scratch_bytes;
  int64_t* scratch_query = ctx.scratch_query;
  if (X6 && !scratch_query) {   // the split path needs room for the split weights; else the caller falls back
    const int64_t need = ((int64_t)3 * weight_taps(p) * p.n_total * (p.c0 + p.c1) * 2 + 255) / 256 * 256;
    if (!scratch || scratch_bytes < need) return ADVOC_ERR_UNSUPPORTED;
  }
  if (name_only) {
    static const std::string name = std::string("gather_gemm_kernel<") + std::to_string(MT) + ", " +
                                    std::to_string(NT) + ", " + std::to_string(WGM) + ", " +
                                    std::to_string(WGN) + ", " + (B_KN ? "true" : "false") + ", " +
                                    std::to_string(BK) + (X6 ? ", true>" : ", false>");
    *name_only = name.c_str();
    return ADVOC_OK;
  }
  const int64_t M = (int64_t)p.batch * p.gh * p.gw;
  if (M > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;       // the kernel splits rows with 32-bit divisions
  const int64_t gx = ceil_div(M, C::BM);
  if (gx * (p.n_total / C::BN) > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
  // Small pixel grids with deep contractions (encoder_5.., decoder_5.. and their gradients) would
  // put < 2 workgroups on a CU, each walking hundreds of K tiles alone: split K until the launch
  // holds ~4 workgroups per CU, keeping >= 16 K tiles per split.
  const int64_t tiles = gx * (p.n_total / C::BN) * p.nphase;
  const int nkt = (p.c0 + p.c1) / BK * p.ntaps;
  int ksplit = 1;
  if (tiles < 512 && split_k_allowed()) {
    ksplit = (int)ceil_div((int64_t)1024, tiles);
    if (ksplit > nkt / 16) ksplit = nkt / 16;
    if (ksplit > 16) ksplit = 16;
    if (ksplit < 1) ksplit = 1;
  }
  TailPlan tail;
  if (ksplit == 1) tail = plan_tail(tiles, nkt);
  const int64_t tail_bytes = (int64_t)sizeof(float) * tail.rem * tail.split * C::BM * C::BN;
  GatherGemmParams px = p;
  if (X6) {
    const int taps = weight_taps(p), ktot = p.c0 + p.c1;
    const int64_t wq_bytes = ((int64_t)3 * taps * p.n_total * ktot * 2 + 255) / 256 * 256;
    if (scratch_query) { *scratch_query = wq_bytes + tail_bytes; return ADVOC_OK; }
    if (!scratch || scratch_bytes < wq_bytes) return ADVOC_ERR_UNSUPPORTED;   // caller falls back to fp32 MFMA
    px.wq = reinterpret_cast<const uint16_t*>(scratch);
    px.wq_taps = taps;
    const int rcw = launch_split_weights(p.w, reinterpret_cast<uint16_t*>(scratch), taps, p.n_total,
                                         p.n_valid ? p.n_valid : p.n_total, ktot, b_kn_src, stream);
    if (rcw != ADVOC_OK) return rcw;
    scratch = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + wq_bytes);
    scratch_bytes -= wq_bytes;
  }
  if (scratch_query) { *scratch_query = tail_bytes; return ADVOC_OK; }
  if (tail.split > 1 && scratch && scratch_bytes >= tail_bytes) {
    GatherGemmParams q = px;
    q.tail_main = tail.main; q.tail_split = tail.split; q.tail_ws = scratch;
    q.tail_cnt = tail_counter_slot();
    if (q.tail_cnt) {
      dim3 grid((unsigned)(tail.main + tail.rem * tail.split), 1, 1);
      auto kern = gather_gemm_kernel<MT, NT, WGM, WGN, B_KN, BK, X6>;
      ADVOC_CLEAR_LAUNCH_ERROR();
      hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, stream, q);
      ADVOC_RETURN_IF_LAUNCH_FAILED();
      return ADVOC_OK;
    }
  }
  if (ksplit > 1) {
    for (int i = 0; i < 2; ++i) {
      const GemmDest& d = p.d[i];
      if (d.p == nullptr || d.accum) continue;
      // logical region only: columns beyond out_w (pitch padding) stay untouched
      hipError_t e = hipMemset2DAsync(d.p, sizeof(float) * (size_t)d.pitch * d.c, 0,
                                      sizeof(float) * (size_t)p.out_w * d.c, (size_t)p.batch * p.out_h, stream);
      if (e != hipSuccess) { note_hip_error(e); return ADVOC_ERR_HIP; }
    }
  }
  dim3 grid((unsigned)(gx * (p.n_total / C::BN)), (unsigned)ksplit, (unsigned)p.nphase);
  auto kern = gather_gemm_kernel<MT, NT, WGM, WGN, B_KN, BK, X6>;
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, stream, px);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

// K tile depth: 32 when every channel slice allows it (fewer barriers, full 128-B rows per
// gather), overridable with ADVOC_IGEMM_BK=16|32 for A/B measurements.
int preferred_bk() { return tuning().igemm_bk; }

// Tile choice (measured on MI355X, AdVoc layer shapes): 128x128 / 128x64 tiles reach ~108 / ~97
// TFLOP/s once >= ~900 workgroups are in the launch (4 resident per CU); below that the chip is
// under-filled (528 workgroups: ~70 TFLOP/s) and 64x64 tiles, which quarter the tile and double
// the residency, win (~80-94 TFLOP/s on the same layers).  32 output channels: 128x32.
template <bool B_KN, int BK>
int dispatch_bk(const GatherGemmParams& p, const LaunchCtx& ctx) {
  const int N = p.n_total;
  const int64_t M = (int64_t)p.batch * p.gh * p.gw;
  // 32 output channels: 128 x 32 (measured 3-11 % faster than 256 x 32 on every such layer: 6 instead
  // of 3 workgroups per CU)
  if (N % 64 != 0) {
    if (BK == 16 && x6_allowed()) {
      const int64_t t32 = ceil_div((int64_t)p.batch * p.gh * p.gw, 128) * p.nphase * (N / 32);
      if (tuning().igemm_x6_n32 != 0 && t32 >= 1024) {
        const int rc = launch_cfg<1, 1, 4, 1, false, 16, true>(p, ctx, B_KN);
        if (rc != ADVOC_ERR_UNSUPPORTED) return rc;
      }
    }
    return launch_cfg<1, 1, 4, 1, B_KN, BK>(p, ctx);
  }
  const int bn = N % 128 == 0 ? 128 : 64;
  if (BK == 16 && x6_allowed()) {
    // split-bf16 path (weights pre-split, so one kernel serves both weight layouts)
    // Measured per layer (tools/layer_times.py, both models): the split path wins from ~500 tiles up
    // (2 per CU: its workgroups are 1.3-1.5x faster than the fp32 ones); 128x128 when that many exist,
    // else 128x64; below, the fp32 kernels with their smaller tiles / split-K stay ahead.
    const int xt = tuning().igemm_x6_tile;
    const int64_t rows128 = ceil_div((int64_t)p.batch * p.gh * p.gw, 128) * p.nphase;
    const int64_t t128 = bn == 128 ? rows128 * (N / 128) : 0, t64 = rows128 * (N / 64);
    int rc = ADVOC_ERR_UNSUPPORTED;
    if (xt == 1) rc = launch_cfg<1, 1, 2, 2, false, 16, true>(p, ctx, B_KN);
    else if (xt == 2) rc = launch_cfg<2, 1, 2, 2, false, 16, true>(p, ctx, B_KN);
    else if (xt == 3 && bn == 128) rc = launch_cfg<2, 2, 2, 2, false, 16, true>(p, ctx, B_KN);
    else if (N % 256 == 0 && x6_wide(rows128 * (N / 256), p.ntaps * (p.c0 + p.c1), p.nphase))
      rc = launch_cfg<2, 4, 2, 2, false, 16, true>(p, ctx, B_KN);
    else if (t128 >= 448) rc = launch_cfg<2, 2, 2, 2, false, 16, true>(p, ctx, B_KN);
    else if (t64 >= 448) rc = launch_cfg<2, 1, 2, 2, false, 16, true>(p, ctx, B_KN);
    if (rc != ADVOC_ERR_UNSUPPORTED) return rc;     // small grid, or no / too little workspace: fp32 MFMA below
  }
  const int64_t big_blocks = ceil_div(M, 128) * (N / bn) * p.nphase;
  // ... and the big tiles only pay off on deep contractions: with taps x channels < 2048 (< 1024
  // for a plain forward epilogue) the 64x64 kernel measured 5-15 % faster on every such layer.
  const int k_total = p.ntaps * (p.c0 + p.c1);
  const bool heavy_epilogue = p.grad_act != ADVOC_ACT_NONE || p.d[1].p != nullptr;
  const bool deep = k_total >= (heavy_epilogue ? 2048 : 1024);
  {
    const int force = tuning().igemm_tile;
    if (force == 1) return launch_cfg<1, 1, 2, 2, B_KN, BK>(p, ctx);
    if (force == 2) return launch_cfg<2, 1, 2, 2, B_KN, BK>(p, ctx);
    if (force == 3 && bn == 128) return launch_cfg<2, 2, 2, 2, B_KN, BK>(p, ctx);
    if (force == 3) return launch_cfg<2, 1, 2, 2, B_KN, BK>(p, ctx);
    if (force == 4 && bn == 128) return launch_cfg<1, 2, 2, 2, B_KN, BK>(p, ctx);   // 64 x 128
    if (force == 4) return launch_cfg<1, 1, 2, 2, B_KN, BK>(p, ctx);
  }
  if (big_blocks < 900 || !deep) return launch_cfg<1, 1, 2, 2, B_KN, BK>(p, ctx);   // 64 x 64
  if (bn == 128) return launch_cfg<2, 2, 2, 2, B_KN, BK>(p, ctx);          // 128 x 128
  return launch_cfg<2, 1, 2, 2, B_KN, BK>(p, ctx);                          // 128 x 64
}

template <bool B_KN>
int dispatch(const GatherGemmParams& p, const LaunchCtx& ctx) {
  const int ktot = p.c0 + p.c1;
  const bool can32 = ktot % 32 == 0 && p.c0 % 32 == 0;
  int bk = preferred_bk();
  if (bk != 16 && bk != 32) bk = 16;
  if (bk == 32 && can32) return dispatch_bk<B_KN, 32>(p, ctx);
  return dispatch_bk<B_KN, 16>(p, ctx);
}

}  // namespace

int launch_gather_gemm(const GatherGemmParams& p, bool b_kn, hipStream_t stream, const char** name_only,
                       float* scratch, int64_t scratch_bytes, int64_t* scratch_query) {
  if (scratch_query) *scratch_query = 0;
  const int ktot = p.c0 + p.c1;
  if (p.batch <= 0 || p.gh <= 0 || p.gw <= 0 || ktot <= 0 || p.n_total <= 0) return ADVOC_ERR_BAD_SHAPE;
  if (ktot % 16 || p.c0 % 16 || p.n_total % 32 || p.n_split % 32) return ADVOC_ERR_UNSUPPORTED;
  if (p.nphase < 1 || p.nphase > kMaxPhases || p.ntaps < 1 || p.ntaps > kMaxTaps) return ADVOC_ERR_UNSUPPORTED;
  // the kernel indexes every tensor with 32-bit element offsets
  const int64_t lim = 0x7fffffffLL;
  if ((int64_t)p.batch * p.a_h * p.a0_pitch * p.c0 > lim || (int64_t)p.batch * p.a_h * p.a1_pitch * p.c1 > lim ||
      (int64_t)p.batch * p.out_h * p.d[0].pitch * p.d[0].c > lim ||
      (int64_t)p.batch * p.out_h * p.d[1].pitch * p.d[1].c > lim)
    return ADVOC_ERR_UNSUPPORTED;
  // K order (L2-miss traffic, rocprofv3 FETCH_SIZE; the run time does not depend on it): taps-inner is the
  // better order almost everywhere (D layer_4 backward-data 1039 -> 117 MB raw, decoder_3 forward 442 -> 122,
  // encoder_3 both directions), except the stride-2 gathers with 32 input channels (one 128-byte line per
  // pixel: encoder_2 / layer_2 forward, 304 vs 175 MB), where walking both 64-byte halves of a line back to
  // back wins.  ADVOC_IGEMM_KORDER=0|1 forces one order for A/B runs.
  const int k_order = tuning().igemm_korder >= 0 ? tuning().igemm_korder
                                                 : ((p.sy == 2 && p.nphase == 1 && ktot <= 32) ? 0 : 1);
  GatherGemmParams q = p;
  q.k_order = k_order;
  q.tail_main = 0; q.tail_split = 0; q.tail_ws = nullptr; q.tail_cnt = nullptr;
  const LaunchCtx ctx = {stream, name_only, scratch, scratch_bytes, scratch_query};
  return b_kn ? dispatch<true>(q, ctx) : dispatch<false>(q, ctx);
}

}  // namespace advoc
