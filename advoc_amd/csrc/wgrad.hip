// Weight / bias gradient kernels (gfx950).
//
// Replaces TF Conv2DBackpropFilter and BiasAddGrad for every layer built at
// models/advoc/advoc_model.py:25-69.   dw[tap][a][b] = sum_g P[g*s + d(tap)][a] * Q[g][b]
// (conv_internal.h): a GEMM per tap whose reduction axis is the pixel grid.
//
//   * wgrad_mfma_kernel: both operands >= 32 channels.  Block tile (32 MT WGM) x (32 NT WGN)
//     channels, K step = 16 grid points, exact-fp32 v_mfma_f32_32x32x2_f32.  Both operand tiles
//     land in LDS as [pixel][channel] straight from coalesced float4 channel loads (the layout
//     the MFMA operands want: lane = channel, register = pixel).  The pixel axis is split over
//     blockIdx.z; partial tiles are combined with hardware fp32 atomics into the zeroed dw.
//   * wgrad_thin_kernel: the gathered operand has <= 2 channels (encoder_1, layer_1, decoder_1,
//     layer_5): HBM-bound outer products, threads own channels of the wide operand.
//   * bias_grad_kernel: per-channel column sums.
#include <stdlib.h>

#include <string>

#include "conv_internal.h"
#include "tuning.h"

namespace advoc {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == ADVOC_ACT_LRELU02) return fmaxf(0.2f * v, v);
  if (act == ADVOC_ACT_RELU) return fmaxf(v, 0.f);
  return v;
}

// float4 of `op` at (img, y, x), channels ch..ch+3 of the concatenated view, transformed.
__device__ __forceinline__ float4 load_op4(const Operand& op, int img, int y, int x, int ch) {
  const bool second = ch >= op.c0;
  const float* src = second ? op.p1 : op.p0;
  const int cs = second ? op.c1 : op.c0;
  const int pitch = second ? op.pitch1 : op.pitch0;
  const int64_t off = (((int64_t)img * op.h + y) * pitch + x) * cs + (second ? ch - op.c0 : ch);
  float4 v = *reinterpret_cast<const float4*>(src + off);
  if (op.scale) {
    const float4 sc = *reinterpret_cast<const float4*>(op.scale + ch);
    const float4 sh = *reinterpret_cast<const float4*>(op.shift + ch);
    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
  }
  v.x = act_fwd(v.x, op.act); v.y = act_fwd(v.y, op.act);
  v.z = act_fwd(v.z, op.act); v.w = act_fwd(v.w, op.act);
  if (op.mask && !second) {
    const uchar4 mk = *reinterpret_cast<const uchar4*>(op.mask + off);
    v.x *= mk.x * op.mask_scale; v.y *= mk.y * op.mask_scale;
    v.z *= mk.z * op.mask_scale; v.w *= mk.w * op.mask_scale;
  }
  return v;
}

__device__ __forceinline__ float load_op1(const Operand& op, int img, int y, int x, int ch) {
  const bool second = ch >= op.c0;
  const float* src = second ? op.p1 : op.p0;
  const int cs = second ? op.c1 : op.c0;
  const int pitch = second ? op.pitch1 : op.pitch0;
  const int64_t off = (((int64_t)img * op.h + y) * pitch + x) * cs + (second ? ch - op.c0 : ch);
  float v = src[off];
  if (op.scale) v = v * op.scale[ch] + op.shift[ch];
  v = act_fwd(v, op.act);
  if (op.mask && !second) v *= op.mask[off] * op.mask_scale;
  return v;
}

// ---------------------------------------------------------------------------------------------
// MFMA weight gradient
// ---------------------------------------------------------------------------------------------
constexpr int WK = 16;  // grid points per K step

template <int MT, int NT, int WGM, int WGN, bool X6 = false>
struct WCfg {
  static constexpr int BM = 32 * MT * WGM;   // channels of P per block
  static constexpr int BN = 32 * NT * WGN;   // channels of Q per block
  static constexpr int LDP = BM + 4;
  static constexpr int LDQ = BN + 4;
  // split-bf16 path: always two slots per thread (one channel quad at two consecutive grid points);
  // with a 64-channel Q tile only half of the threads carry Q slots
  static constexpr int P_LOADS = X6 ? 2 : (WK * BM / 4 + 255) / 256;
  static constexpr int Q_LOADS = X6 ? 2 : (WK * BN / 4 + 255) / 256;
  // split-bf16 path (see igemm.hip): per operand three planes of [8 pixel pairs][channels] dwords, one dword =
  // the bf16 terms of two consecutive grid points of one channel
  static constexpr int XP = 3 * (WK / 2) * LDP, XQ = 3 * (WK / 2) * LDQ;   // dwords per stage
  static constexpr size_t LDS_BYTES = X6 ? sizeof(float) * 2 * (XP + XQ) : sizeof(float) * 2 * (WK * LDP + WK * LDQ);
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// x = x0 + x1 + x2 exactly, each a bf16 by truncation (the high halves of h0, h1, h2); see igemm.hip
__device__ __forceinline__ void split3(float x, unsigned& h0, unsigned& h1, unsigned& h2) {
  h0 = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(h0);
  h1 = __float_as_uint(r1) & 0xffff0000u;
  h2 = __float_as_uint(r1 - __uint_as_float(h1));
}
__device__ __forceinline__ unsigned pack_hi16(unsigned lo, unsigned hi) {
  return __builtin_amdgcn_perm(hi, lo, 0x07060302u);
}
// the three planes' dwords for one channel at two consecutive grid points (a = even, b = odd)
__device__ __forceinline__ void split_pair(float a, float b, unsigned& d0, unsigned& d1, unsigned& d2) {
  unsigned a0, a1, a2, b0, b1, b2;
  split3(a, a0, a1, a2);
  split3(b, b0, b1, b2);
  d0 = pack_hi16(a0, b0); d1 = pack_hi16(a1, b1); d2 = pack_hi16(a2, b2);
}

// float4 of `op` at a precomputed pixel index (img*h + y)*pitch + x per source, channels ch..ch+3
__device__ __forceinline__ float4 load_op4_at(const Operand& op, int pix0, int pix1, int ch, float slope) {
  const bool second = ch >= op.c0;
  const float* src = second ? op.p1 : op.p0;
  const int off = second ? pix1 * op.c1 + (ch - op.c0) : pix0 * op.c0 + ch;
  float4 v = *reinterpret_cast<const float4*>(src + off);
  if (op.scale) {
    const float4 sc = *reinterpret_cast<const float4*>(op.scale + ch);
    const float4 sh = *reinterpret_cast<const float4*>(op.shift + ch);
    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
  }
  v.x = fmaxf(v.x, slope * v.x); v.y = fmaxf(v.y, slope * v.y);
  v.z = fmaxf(v.z, slope * v.z); v.w = fmaxf(v.w, slope * v.w);
  if (op.mask && !second) {   // the mask covers source 0 only
    const uchar4 mk = *reinterpret_cast<const uchar4*>(op.mask + off);
    v.x *= mk.x * op.mask_scale; v.y *= mk.y * op.mask_scale;
    v.z *= mk.z * op.mask_scale; v.w *= mk.w * op.mask_scale;
  }
  return v;
}

// transforms are applied at the LDS-store stage, so that the raw load itself can stay in flight
// across the MFMA block
__device__ __forceinline__ float4 transform4(const Operand& op, float4 v, int ch, int off, float slope, bool live) {
  if (op.scale) {
    const float4 sc = *reinterpret_cast<const float4*>(op.scale + ch);
    const float4 sh = *reinterpret_cast<const float4*>(op.shift + ch);
    const float k = live ? 1.f : 0.f;     // padding stays zero
    v.x = v.x * sc.x + sh.x * k; v.y = v.y * sc.y + sh.y * k; v.z = v.z * sc.z + sh.z * k; v.w = v.w * sc.w + sh.w * k;
  }
  if (slope != 1.f) {     // uniform per operand: the dY operand has no activation
    v.x = fmaxf(v.x, slope * v.x); v.y = fmaxf(v.y, slope * v.y);
    v.z = fmaxf(v.z, slope * v.z); v.w = fmaxf(v.w, slope * v.w);
  }
  if (op.mask && ch < op.c0 && live) {
    const uchar4 mk = *reinterpret_cast<const uchar4*>(op.mask + off);
    v.x *= mk.x * op.mask_scale; v.y *= mk.y * op.mask_scale;
    v.z *= mk.z * op.mask_scale; v.w *= mk.w * op.mask_scale;
  }
  return v;
}

__device__ __forceinline__ float slope_of(int act) {
  return act == ADVOC_ACT_LRELU02 ? 0.2f : (act == ADVOC_ACT_RELU ? 0.f : 1.f);
}

// launch_bounds(256, 2): see igemm.hip -- keeps the prefetch registers out of scratch.
template <int MT, int NT, int WGM, int WGN, bool X6 = false>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(const WgradParams p, int tiles_n, int tiles,
                                                            int chunk) {
  using C = WCfg<MT, NT, WGM, WGN, X6>;
  static_assert(!X6 || (C::BM == 128 && (C::BN == 128 || C::BN == 64)), "split-bf16 path: 128 x 128 / 128 x 64 channel tiles");
  constexpr int BM = C::BM, BN = C::BN;
  constexpr int PL = C::P_LOADS, QL = C::Q_LOADS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ps = smem;                      // [2][WK][LDP]
  float* Qs = smem + 2 * WK * C::LDP;    // [2][WK][LDQ]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int half = lane >> 5, l32 = lane & 31;
  // XCD-aware block order.  Workgroup b runs on XCD b % 8, each with its own L2; the `tiles` blocks
  // of one pixel chunk read the SAME P and Q pixels (different taps / channel tiles), so an XCD
  // walks a contiguous run of (chunk, tile) pairs and those re-reads hit its L2 instead of being
  // fetched once per XCD.
  int vblock;
  {
    const int nb = gridDim.x, b = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = b & 7, slot = b >> 3;
    vblock = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;   // bijective for any nb
  }
  const int zchunk = vblock / tiles;
  const int tile_id = vblock - zchunk * tiles;
  // rows of the GEMM = (tap, channel of P): every tap shares the Q tile of a grid point
  const int a0 = (tile_id / tiles_n) * BM;
  const int b0 = (tile_id % tiles_n) * BN;
  const int ca = p.P.c0 + p.P.c1, cb = p.Q.c0 + p.Q.c1;
  const int rows_total = p.ntaps * ca;
  const int64_t M = (int64_t)p.batch * p.gh * p.gw;
  const int64_t g_begin = (int64_t)zchunk * chunk;
  const int64_t g_end = g_begin + chunk < M ? g_begin + chunk : M;
  const int nkt = (int)((g_end - g_begin + WK - 1) / WK);
  const float pslope = slope_of(p.P.act), qslope = slope_of(p.Q.act);

  // Loader slots.  Slot i of the P (Q) tile = pixel k_i of the K step, channel quad cq_i; the
  // slot's grid point advances by WK per K step and is tracked incrementally (no divisions).
  // (slot -> tile position is recomputed from tid where needed: registers are what keeps this
  //  kernel at 3 workgroups per CU instead of 4)
  // split-bf16 path: a thread's two slots are the SAME channel quad at two consecutive grid points
  // (2 kp, 2 kp + 1), so that their bf16 terms pack into one dword per channel and plane
#define P_K(i) (X6 ? 2 * (tid / (BM / 4)) + (i) : (tid + 256 * (i)) / (BM / 4))
#define P_COL(i) (X6 ? 4 * (tid % (BM / 4)) : 4 * ((tid + 256 * (i)) % (BM / 4)))
#define Q_K(i) (X6 ? 2 * (tid / (BN / 4)) + (i) : (tid + 256 * (i)) / (BN / 4))
#define Q_COL(i) (X6 ? 4 * (tid % (BN / 4)) : 4 * ((tid + 256 * (i)) % (BN / 4)))
  // Each slot keeps the ELEMENT OFFSET of its current pixel in its source tensor (a slot never
  // changes source: its channel quad is fixed) and moves it with three per-slot deltas -- 16 grid
  // points along x, x wrap to the next grid row, y wrap to the next image -- instead of rebuilding
  // ((img * H + y) * pitch + x) * C every K step (four 32-bit multiply-adds per slot and step,
  // quarter-rate VALU work that competed with the MFMAs).
  int p_ch[PL], p_gx[PL], p_gy[PL], p_x[PL], p_y[PL], p_off[PL], p_ds[PL], p_dwx[PL], p_dwy[PL];
  int q_gx[QL], q_gy[QL], q_off[QL], q_ds[QL], q_dwx[QL], q_dwy[QL];
  bool p_on[PL], q_on[QL], p_second[PL], q_second[QL];
#pragma unroll
  for (int i = 0; i < PL; ++i) {
    const int row = a0 + P_COL(i);
    p_on[i] = P_K(i) < WK && row < rows_total;
    const int t_ = p_on[i] ? row / ca : 0;
    p_ch[i] = row - t_ * ca;
    const int tw = p.tap[t_];
    const int64_t g = g_begin + (P_K(i) < WK ? P_K(i) : 0);
    p_gx[i] = (int)(g % p.gw);
    const int64_t t = g / p.gw;
    p_gy[i] = (int)(t % p.gh);
    const int img = (int)(t / p.gh);
    p_second[i] = p_ch[i] >= p.P.c0;
    const int cs = p_second[i] ? p.P.c1 : p.P.c0, pitch = p_second[i] ? p.P.pitch1 : p.P.pitch0;
    p_x[i] = p_gx[i] * p.sx + (int)(int8_t)((tw >> 8) & 0xff);
    p_y[i] = p_gy[i] * p.sy + (int)(int8_t)(tw & 0xff);
    p_off[i] = (int)((((int64_t)img * p.P.h + p_y[i]) * pitch + p_x[i]) * cs + (p_second[i] ? p_ch[i] - p.P.c0 : p_ch[i]));
    p_ds[i] = WK * p.sx * cs;
    p_dwx[i] = (p.sy * pitch - p.gw * p.sx) * cs;
    p_dwy[i] = (p.P.h - p.gh * p.sy) * pitch * cs;
  }
#pragma unroll
  for (int i = 0; i < QL; ++i) {
    const int ch = b0 + Q_COL(i);
    q_on[i] = Q_K(i) < WK && ch < cb;
    const int64_t g = g_begin + (Q_K(i) < WK ? Q_K(i) : 0);
    q_gx[i] = (int)(g % p.gw);
    const int64_t t = g / p.gw;
    q_gy[i] = (int)(t % p.gh);
    const int img = (int)(t / p.gh);
    q_second[i] = ch >= p.Q.c0;
    const int cs = q_second[i] ? p.Q.c1 : p.Q.c0, pitch = q_second[i] ? p.Q.pitch1 : p.Q.pitch0;
    q_off[i] = (int)((((int64_t)img * p.Q.h + q_gy[i]) * pitch + q_gx[i]) * cs + (q_second[i] ? ch - p.Q.c0 : ch));
    q_ds[i] = WK * cs;
    q_dwx[i] = (pitch - p.gw) * cs;
    q_dwy[i] = (p.Q.h - p.gh) * pitch * cs;
  }
  const int span = (int)(g_end - g_begin);   // grid points of this chunk (chunk fits 31 bits)

  float4 rp[PL], rq[QL];
  int rp_off[PL], rq_off[QL];
  unsigned rp_live = 0, rq_live = 0;

  // loads the slots' current grid points, then advances them by WK
#define ADVOC_W_LOAD(KT)                                                                             \
  {                                                                                                  \
    const int kb_ = (KT) * WK;                                                                       \
    rp_live = 0; rq_live = 0;                                                                        \
    _Pragma("unroll") for (int i = 0; i < PL; ++i) {                                                 \
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                    \
      rp_off[i] = 0;                                                                                 \
      if (p_on[i] && kb_ + P_K(i) < span && (unsigned)p_y[i] < (unsigned)p.P.h &&                    \
          (unsigned)p_x[i] < (unsigned)p.P.w) {                                                      \
        v = *reinterpret_cast<const float4*>((p_second[i] ? p.P.p1 : p.P.p0) + p_off[i]);           \
        rp_off[i] = p_off[i];                                                                        \
        rp_live |= 1u << i;                                                                          \
      }                                                                                              \
      rp[i] = v;                                                                                     \
      p_gx[i] += WK; p_x[i] += WK * p.sx; p_off[i] += p_ds[i];                                       \
      while (p_gx[i] >= p.gw) {                                                                      \
        p_gx[i] -= p.gw; p_x[i] -= p.gw * p.sx; p_y[i] += p.sy; p_off[i] += p_dwx[i];                \
        if (++p_gy[i] >= p.gh) { p_gy[i] = 0; p_y[i] -= p.gh * p.sy; p_off[i] += p_dwy[i]; }         \
      }                                                                                              \
    }                                                                                                \
    _Pragma("unroll") for (int i = 0; i < QL; ++i) {                                                 \
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                    \
      rq_off[i] = 0;                                                                                 \
      if (q_on[i] && kb_ + Q_K(i) < span) {                                                          \
        v = *reinterpret_cast<const float4*>((q_second[i] ? p.Q.p1 : p.Q.p0) + q_off[i]);           \
        rq_off[i] = q_off[i];                                                                        \
        rq_live |= 1u << i;                                                                          \
      }                                                                                              \
      rq[i] = v;                                                                                     \
      q_gx[i] += WK; q_off[i] += q_ds[i];                                                            \
      while (q_gx[i] >= p.gw) {                                                                      \
        q_gx[i] -= p.gw; q_off[i] += q_dwx[i];                                                       \
        if (++q_gy[i] >= p.gh) { q_gy[i] = 0; q_off[i] += q_dwy[i]; }                                \
      }                                                                                              \
    }                                                                                                \
  }

#define ADVOC_W_STORE(BUF)                                                                           \
  {                                                                                                  \
    if constexpr (X6) {                                                                              \
      unsigned* Px_ = reinterpret_cast<unsigned*>(smem) + (BUF) * (C::XP + C::XQ);                   \
      unsigned* Qx_ = Px_ + C::XP;                                                                   \
      const int kp_ = tid / (BM / 4);                                                                \
      {                                                                                              \
        const float4 e_ = transform4(p.P, rp[0], p_ch[0], rp_off[0], pslope, rp_live & 1u);          \
        const float4 o_ = transform4(p.P, rp[1], p_ch[1], rp_off[1], pslope, (rp_live >> 1) & 1u);   \
        uint4 d0_, d1_, d2_;                                                                         \
        split_pair(e_.x, o_.x, d0_.x, d1_.x, d2_.x); split_pair(e_.y, o_.y, d0_.y, d1_.y, d2_.y);    \
        split_pair(e_.z, o_.z, d0_.z, d1_.z, d2_.z); split_pair(e_.w, o_.w, d0_.w, d1_.w, d2_.w);    \
        unsigned* w_ = Px_ + kp_ * C::LDP + P_COL(0);                                                \
        *reinterpret_cast<uint4*>(w_) = d0_;                                                         \
        *reinterpret_cast<uint4*>(w_ + (WK / 2) * C::LDP) = d1_;                                     \
        *reinterpret_cast<uint4*>(w_ + 2 * (WK / 2) * C::LDP) = d2_;                                 \
      }                                                                                              \
      {                                                                                              \
        const int kq_ = tid / (BN / 4);                                                              \
        const float4 e_ = transform4(p.Q, rq[0], b0 + Q_COL(0), rq_off[0], qslope, rq_live & 1u);    \
        const float4 o_ = transform4(p.Q, rq[1], b0 + Q_COL(1), rq_off[1], qslope, (rq_live >> 1) & 1u); \
        uint4 d0_, d1_, d2_;                                                                         \
        split_pair(e_.x, o_.x, d0_.x, d1_.x, d2_.x); split_pair(e_.y, o_.y, d0_.y, d1_.y, d2_.y);    \
        split_pair(e_.z, o_.z, d0_.z, d1_.z, d2_.z); split_pair(e_.w, o_.w, d0_.w, d1_.w, d2_.w);    \
        unsigned* w_ = Qx_ + kq_ * C::LDQ + Q_COL(0);                                                \
        if (kq_ < WK / 2) {                                                                          \
          *reinterpret_cast<uint4*>(w_) = d0_;                                                       \
          *reinterpret_cast<uint4*>(w_ + (WK / 2) * C::LDQ) = d1_;                                   \
          *reinterpret_cast<uint4*>(w_ + 2 * (WK / 2) * C::LDQ) = d2_;                               \
        }                                                                                            \
      }                                                                                              \
      (void)kp_;                                                                                     \
    } else {                                                                                         \
    float* Pb_ = Ps + (BUF) * WK * C::LDP;                                                           \
    float* Qb_ = Qs + (BUF) * WK * C::LDQ;                                                           \
    _Pragma("unroll") for (int i = 0; i < PL; ++i)                                                   \
        if (P_K(i) < WK) *reinterpret_cast<float4*>(Pb_ + P_K(i) * C::LDP + P_COL(i)) =              \
            transform4(p.P, rp[i], p_ch[i], rp_off[i], pslope, (rp_live >> i) & 1u);                 \
    _Pragma("unroll") for (int i = 0; i < QL; ++i)                                                   \
        if (Q_K(i) < WK) *reinterpret_cast<float4*>(Qb_ + Q_K(i) * C::LDQ + Q_COL(i)) =              \
            transform4(p.Q, rq[i], b0 + Q_COL(i), rq_off[i], qslope, (rq_live >> i) & 1u);           \
    }                                                                                                \
  }

  floatx16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nkt == 0) return;
  ADVOC_W_LOAD(0);
  ADVOC_W_STORE(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    // unconditional prefetch; past the end every slot fails the g_end test and loads nothing
    ADVOC_W_LOAD(kt + 1);
    if constexpr (X6) {
      // lane (l32, half): grid points [8 half, 8 half + 8) = pixel pairs 4 half .. 4 half + 3 of its channel;
      // six bf16 products per 32x32x16 block, smallest terms first (igemm.hip)
      const unsigned* Px = reinterpret_cast<const unsigned*>(smem) + buf * (C::XP + C::XQ);
      const unsigned* Qx = Px + C::XP;
      bf16x8 af[MT][3], bq[NT][3];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const unsigned* q = Px + (pl * (WK / 2) + 4 * half) * C::LDP + (wm * MT + i) * 32 + l32;
          const uint4 d = make_uint4(q[0], q[C::LDP], q[2 * C::LDP], q[3 * C::LDP]);
          af[i][pl] = __builtin_bit_cast(bf16x8, d);
        }
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const unsigned* q = Qx + (pl * (WK / 2) + 4 * half) * C::LDQ + (wn * NT + j) * 32 + l32;
          const uint4 d = make_uint4(q[0], q[C::LDQ], q[2 * C::LDQ], q[3 * C::LDQ]);
          bq[j][pl] = __builtin_bit_cast(bf16x8, d);
        }
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
    const float* Pb = Ps + buf * WK * C::LDP;
    const float* Qb = Qs + buf * WK * C::LDQ;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float a[MT], b[NT];
      const int k = 2 * s + half;   // MFMA K slot: lanes 0-31 -> k, lanes 32-63 -> k+1
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = Pb[k * C::LDP + (wm * MT + i) * 32 + l32];
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = Qb[k * C::LDQ + (wn * NT + j) * 32 + l32];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    }
    if (kt + 1 < nkt) ADVOC_W_STORE(buf ^ 1);
    __syncthreads();
  }
#undef ADVOC_W_LOAD
#undef ADVOC_W_STORE
#undef P_K
#undef P_COL
#undef Q_K
#undef Q_COL

#pragma unroll
  for (int i = 0; i < MT; ++i) {
    // a 32-row MFMA tile never straddles a tap (ca % 32 == 0): one tap lookup per tile
    const int row0 = a0 + (wm * MT + i) * 32;
    if (row0 >= rows_total) continue;
    const int t_ = row0 / ca;
    float* out = p.dw + ((int64_t)(p.tap[t_] >> 16) * ca + (row0 - t_ * ca)) * cb;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int b = b0 + (wn * NT + j) * 32 + l32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int a = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (b < cb) unsafeAtomicAdd(out + (int64_t)a * cb + b, acc[i][j][r]);
      }
    }
  }
}

template <int MT, int NT, int WGM, int WGN, bool X6 = false>
int launch_wcfg(const WgradParams& p, hipStream_t stream, const char** name_only) {
  using C = WCfg<MT, NT, WGM, WGN, X6>;
  if (name_only) {
    static const std::string name = std::string("wgrad_mfma_kernel<") + std::to_string(MT) + ", " +
                                    std::to_string(NT) + ", " + std::to_string(WGM) + ", " +
                                    std::to_string(WGN) + (X6 ? ", true>" : ", false>");
    *name_only = name.c_str();
    return ADVOC_OK;
  }
  if (!p.accumulate) {
    const int ca_ = p.P.c0 + p.P.c1, cb_ = p.Q.c0 + p.Q.c1;
    hipError_t e = hipMemsetAsync(p.dw, 0, sizeof(float) * (size_t)p.ntaps * ca_ * cb_, stream);
    if (e != hipSuccess) { note_hip_error(e); return ADVOC_ERR_HIP; }
  }
  const int ca = p.P.c0 + p.P.c1, cb = p.Q.c0 + p.Q.c1;
  const int tiles_m = (p.ntaps * ca + C::BM - 1) / C::BM, tiles_n = (cb + C::BN - 1) / C::BN;
  const int64_t M = (int64_t)p.batch * p.gh * p.gw;
  // Split the pixel axis so the launch is whole rounds of the chip: `resident` workgroups fit at
  // once (occupancy query, cached per kernel instance); a launch of 1.33 x resident would run a
  // 1/3-full second round.  Two rounds when the work allows it (shorter tail, same traffic).
  static const int64_t resident = [] {
    int per_cu = 0, dev = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wgrad_mfma_kernel<MT, NT, WGM, WGN, X6>, 256,
                                                     C::LDS_BYTES) != hipSuccess || per_cu < 1)
      per_cu = 3;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    (void)hipGetLastError();
    return (int64_t)per_cu * cus;
  }();
  const int64_t tiles = (int64_t)tiles_m * tiles_n;
  int64_t ksplit = ceil_div(resident, tiles);
  if (M / ksplit >= 4096) ksplit = ceil_div(2 * resident, tiles);
  const int64_t max_split = ceil_div(M, 8 * WK);
  if (ksplit > max_split) ksplit = max_split;
  if (ksplit < 1) ksplit = 1;
  int64_t chunk = ceil_div(ceil_div(M, ksplit), WK) * WK;
  ksplit = ceil_div(M, chunk);
  if (chunk > 0x7fffffffLL || tiles * ksplit > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
  dim3 grid((unsigned)(tiles * ksplit), 1, 1);
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL((wgrad_mfma_kernel<MT, NT, WGM, WGN, X6>), grid, dim3(256), C::LDS_BYTES, stream, p,
                     tiles_n, (int)tiles, (int)chunk);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

// ---------------------------------------------------------------------------------------------
// thin weight gradient: P has <= 2 channels.  Thread = one channel b of Q; a block walks a
// chunk of grid points; per point the (<= 2 x ntaps) P scalars are wave-uniform broadcasts.
// ---------------------------------------------------------------------------------------------
template <int CA>
__global__ __launch_bounds__(256) void wgrad_thin_kernel(const WgradParams p, int chunk) {
  __shared__ float red[256];
  const int cb = p.Q.c0 + p.Q.c1;
  // threads: tb = channel lane within a 64-wide channel group, tg = pixel sub-group
  const int lanes_b = cb < 64 ? cb : 64;            // cb is a multiple of 32
  const int groups = 256 / lanes_b;
  const int tb = threadIdx.x % lanes_b, tg = threadIdx.x / lanes_b;
  const int b = blockIdx.y * lanes_b + tb;
  const int64_t M = (int64_t)p.batch * p.gh * p.gw;
  const int64_t g_begin = (int64_t)blockIdx.x * chunk;
  const int64_t g_end = g_begin + chunk < M ? g_begin + chunk : M;

  float acc[kMaxTaps][CA];
#pragma unroll
  for (int t = 0; t < kMaxTaps; ++t)
#pragma unroll
    for (int a = 0; a < CA; ++a) acc[t][a] = 0.f;

  for (int64_t g = g_begin + tg; g < g_end; g += groups) {
    const int gx = (int)(g % p.gw);
    const int64_t tt = g / p.gw;
    const int gy = (int)(tt % p.gh), img = (int)(tt / p.gh);
    const float q = load_op1(p.Q, img, gy, gx, b);
#pragma unroll
    for (int t = 0; t < kMaxTaps; ++t) {
      if (t < p.ntaps) {
        const int tp = p.tap[t];
        const int y = gy * p.sy + (int)(int8_t)(tp & 0xff), x = gx * p.sx + (int)(int8_t)((tp >> 8) & 0xff);
        if ((unsigned)y < (unsigned)p.P.h && (unsigned)x < (unsigned)p.P.w) {
#pragma unroll
          for (int a = 0; a < CA; ++a) acc[t][a] = fmaf(load_op1(p.P, img, y, x, a), q, acc[t][a]);
        }
      }
    }
  }
  // reduce the pixel sub-groups through LDS, one (tap, a) at a time, then one atomic per channel
#pragma unroll
  for (int t = 0; t < kMaxTaps; ++t) {
    if (t >= p.ntaps) break;
#pragma unroll
    for (int a = 0; a < CA; ++a) {
      red[threadIdx.x] = acc[t][a];
      __syncthreads();
      if (tg == 0) {
        float s = 0.f;
        for (int k = 0; k < groups; ++k) s += red[k * lanes_b + tb];
        const int wtap = p.tap[t] >> 16;
        unsafeAtomicAdd(p.dw + ((int64_t)wtap * CA + a) * cb + b, s);
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------
// bias gradient
// ---------------------------------------------------------------------------------------------
// Column sums with 16-byte loads: when c % 4 == 0 and rows are dense a thread owns 4 channels
// and walks pixels; lanes of a wave cover consecutive 16-byte pieces (full 128-B lines).
__global__ __launch_bounds__(256) void bias_grad_vec4_kernel(const float* __restrict__ dy,
                                                             const uint8_t* __restrict__ mask,
                                                             float mask_scale, int64_t npix, int c,
                                                             float* __restrict__ db) {
  __shared__ float4 red[256];
  const int quads = c / 4;                        // float4 columns
  const int lanes_q = quads < 256 ? quads : 256;  // quads is a power-of-two multiple of 8 here or < 256
  const int groups = 256 / lanes_q;
  const int tq = threadIdx.x % lanes_q, tg = threadIdx.x / lanes_q;
  for (int qbase = 0; qbase < quads; qbase += lanes_q) {
    const int q = qbase + tq;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tg < groups && q < quads) {
      const int64_t step = (int64_t)gridDim.x * groups;
      int64_t pix = (int64_t)blockIdx.x * groups + tg;
      // four independent 16-byte loads in flight per lane
      float4 acc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (; pix + 3 * step < npix; pix += 4 * step) {
        float4 v[4];
        int64_t o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          o[u] = (pix + u * step) * c + 4 * q;
          v[u] = *reinterpret_cast<const float4*>(dy + o[u]);
        }
        if (mask) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uchar4 m = *reinterpret_cast<const uchar4*>(mask + o[u]);
            v[u].x *= m.x * mask_scale; v[u].y *= m.y * mask_scale;
            v[u].z *= m.z * mask_scale; v[u].w *= m.w * mask_scale;
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc[u].x += v[u].x; acc[u].y += v[u].y; acc[u].z += v[u].z; acc[u].w += v[u].w;
        }
      }
      for (; pix < npix; pix += step) {
        const int64_t off = pix * c + 4 * q;
        float4 v = *reinterpret_cast<const float4*>(dy + off);
        if (mask) {
          const uchar4 mk = *reinterpret_cast<const uchar4*>(mask + off);
          v.x *= mk.x * mask_scale; v.y *= mk.y * mask_scale; v.z *= mk.z * mask_scale; v.w *= mk.w * mask_scale;
        }
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { s.x += acc[u].x; s.y += acc[u].y; s.z += acc[u].z; s.w += acc[u].w; }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (tg == 0 && q < quads) {
      float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = 0; k < groups; ++k) {
        const float4 r = red[k * lanes_q + tq];
        tot.x += r.x; tot.y += r.y; tot.z += r.z; tot.w += r.w;
      }
      unsafeAtomicAdd(db + 4 * q, tot.x);
      unsafeAtomicAdd(db + 4 * q + 1, tot.y);
      unsafeAtomicAdd(db + 4 * q + 2, tot.z);
      unsafeAtomicAdd(db + 4 * q + 3, tot.w);
    }
    __syncthreads();
  }
}

// generic fallback (any c, pitched rows): thread -> channel x pixel sub-group
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dy,
                                                        const uint8_t* __restrict__ mask,
                                                        float mask_scale, int64_t rows, int w,
                                                        int pitch, int c, float* __restrict__ db) {
  __shared__ float red[256];
  const int lanes_c = c < 256 ? c : 256;
  const int groups = 256 / lanes_c;
  const int tc = threadIdx.x % lanes_c, tg = threadIdx.x / lanes_c;
  const bool idle = tg >= groups;   // 256 % c != 0
  const int64_t npix = rows * w;
  for (int cbase = 0; cbase < c; cbase += lanes_c) {
    const int ch = cbase + tc;
    float s = 0.f;
    if (!idle && ch < c) {
      for (int64_t q = (int64_t)blockIdx.x * groups + tg; q < npix; q += (int64_t)gridDim.x * groups) {
        const int64_t row = q / w;
        const int x = (int)(q - row * w);
        const int64_t off = (row * pitch + x) * c + ch;
        float v = dy[off];
        if (mask) v *= mask[off] * mask_scale;
        s += v;
      }
    }
    red[threadIdx.x] = idle ? 0.f : s;
    __syncthreads();
    if (tg == 0 && ch < c) {
      float tot = 0.f;
      for (int k = 0; k < groups; ++k) tot += red[k * lanes_c + tc];
      unsafeAtomicAdd(db + ch, tot);
    }
    __syncthreads();
  }
}

}  // namespace

int launch_wgrad_mfma(const WgradParams& p, hipStream_t stream, const char** name_only) {
  const int ca = p.P.c0 + p.P.c1, cb = p.Q.c0 + p.Q.c1;
  if (ca % 32 || cb % 32 || p.P.c0 % 4 || p.Q.c0 % 4) return ADVOC_ERR_UNSUPPORTED;
  if (p.P.c1 && p.P.c0 % 32) return ADVOC_ERR_UNSUPPORTED;  // a 32-row MFMA tile stays in one source
  if (p.Q.c1 && p.Q.c0 % 32) return ADVOC_ERR_UNSUPPORTED;
  // rows = taps x ca (>= 512): 128-row tiles; the column tile follows cb
  if (cb % 128 == 0) {                                                       // 128 x 128
    // split-bf16 path (igemm.hip): both operands split into three bf16 terms while they are parked in LDS;
    // ADVOC_WGRAD_X6=0 keeps the fp32 MFMA kernel (A/B measurements)
    if (tuning().wgrad_x6 != 0) return launch_wcfg<2, 2, 2, 2, true>(p, stream, name_only);
    return launch_wcfg<2, 2, 2, 2>(p, stream, name_only);
  }
  if (cb % 64 == 0) {                                                        // 128 x 64
    // the split variant of this tile measured 3-6 % SLOWER than fp32 (half of the threads carry no Q slot):
    // only on request (ADVOC_WGRAD_X6=2)
    if (tuning().wgrad_x6 == 2) return launch_wcfg<2, 1, 2, 2, true>(p, stream, name_only);
    return launch_wcfg<2, 1, 2, 2>(p, stream, name_only);
  }
  return launch_wcfg<1, 1, 4, 1>(p, stream, name_only);                      // 128 x 32
}

int launch_wgrad_thin(const WgradParams& p, hipStream_t stream, const char** name_only) {
  const int ca = p.P.c0 + p.P.c1, cb = p.Q.c0 + p.Q.c1;
  if (ca < 1 || ca > 2 || cb % 32) return ADVOC_ERR_UNSUPPORTED;
  if (name_only) {
    *name_only = ca == 1 ? "wgrad_thin_kernel<1>" : "wgrad_thin_kernel<2>";
    return ADVOC_OK;
  }
  if (!p.accumulate) {
    hipError_t e = hipMemsetAsync(p.dw, 0, sizeof(float) * (size_t)p.ntaps * ca * cb, stream);
    if (e != hipSuccess) { note_hip_error(e); return ADVOC_ERR_HIP; }
  }
  const int64_t M = (int64_t)p.batch * p.gh * p.gw;
  const int lanes_b = cb < 64 ? cb : 64;
  const int by = cb / lanes_b;
  int64_t nblk = 2048 / by;
  if (nblk < 1) nblk = 1;
  int64_t chunk = ceil_div(M, nblk);
  if (chunk < 64) chunk = 64;
  nblk = ceil_div(M, chunk);
  if (chunk > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
  dim3 grid((unsigned)nblk, (unsigned)by);
  ADVOC_CLEAR_LAUNCH_ERROR();
  if (ca == 1) hipLaunchKernelGGL(wgrad_thin_kernel<1>, grid, dim3(256), 0, stream, p, (int)chunk);
  else hipLaunchKernelGGL(wgrad_thin_kernel<2>, grid, dim3(256), 0, stream, p, (int)chunk);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

int launch_bias_grad(const float* dy, const uint8_t* mask, float mask_scale, int64_t rows, int w,
                     int pitch, int c, float* db, int accumulate, hipStream_t stream) {
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(db, 0, sizeof(float) * (size_t)c, stream);
    if (e != hipSuccess) { note_hip_error(e); return ADVOC_ERR_HIP; }
  }
  ADVOC_CLEAR_LAUNCH_ERROR();
  const int quads = c / 4;
  if (c % 4 == 0 && pitch == w && 256 % (quads < 256 ? quads : 256) == 0) {
    const int groups = 256 / (quads < 256 ? quads : 256);
    // ONE workgroup per CU: 256 workgroups already stream at ~5.5 TB/s (tools/micro/colsum_bw.hip),
    // and every extra workgroup costs ~40 ns of serialised same-address atomics at the end
    // (1024 workgroups: 81 us, 256: 49 us for a 268 MB tensor).
    int64_t blocks = ceil_div(rows * w, (int64_t)groups * 16);
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(bias_grad_vec4_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dy, mask,
                       mask_scale, rows * w, c, db);
  } else {
    const int lanes_c = c < 256 ? c : 256;
    const int groups = 256 / lanes_c;
    int64_t blocks = ceil_div(rows * w, (int64_t)groups * 64);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(bias_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dy, mask,
                       mask_scale, rows, w, pitch, c, db);
  }
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

}  // namespace advoc
