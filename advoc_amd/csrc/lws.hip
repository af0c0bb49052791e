// Local Weighted Sums (LWS) phase reconstruction on the GPU: the reference's default vocoder back end,
// lws.lws(nfft, nhop, mode='speech', perfectrec=False).run_lws(X_mag) at advoc/spectral.py:314-326 and
// models/advoc/spectral_util.py:45-50 (third-party lws 1.2, C++, not part of /root/reference: the PUBLISHED algorithm
// is restated -- Le Roux et al., DAFx 2010 / ASJ 2010 -- and the result is parity-UNPINNED; oracle/lws_np.py is the
// CPU restatement these kernels are tested against).
//
// A complex spectrogram X is consistent (the STFT of a signal) iff X = P X with P = STFT o iSTFT, and P is a small
// 2-D convolution: (P X)[t, f] = sum_{q, p} K_q(p) R_q(f + p) X[t + q, f + p], |q| < Q = nfft / nhop, kernel decaying
// fast in |p| (truncated to |p| < L), R_q(m) = exp(-2 pi i m q nhop / nfft) for lws's frame-local phase convention
// (periodic in m with period P = nfft / gcd(nfft, nhop): the caller hands over the products K_q(p) R_q(m mod P)).
// LWS iterates  X[t, f] <- |A[t, f]| phase( sum_{(q, p) != (0, 0)} ... )  in three stages:
//   lws_causal_kernel   one workgroup per clip walks the frames in time order with a 2Q-frame ring in LDS: a new frame
//                       (look-ahead frames ahead) is initialised from the frames before it only ("no future"), the
//                       current frame is refined `online_iterations` times with the look-ahead (Jacobi over its bins);
//   lws_batch_kernel    one sweep over the whole spectrogram (all bins from the previous iterate), touching only bins
//                       above the sweep's magnitude threshold; the host launches `batch_iterations` of them, ping-pong.
// Both are cache / LDS bound: 63 complex taps per bin.
#include <stdlib.h>

#include "common.h"

namespace {

// W[q + Q - 1][p + L - 1][m mod P] = alpha_q(p) * exp(-2 pi i m q nhop / nfft), m = f + p: the rotation only depends on
// m modulo P = nfft / gcd(nfft, nhop) (4 for the reference's nfft 1024 / hop 256), so the whole kernel is a table of
// (2Q - 1)(2L - 1) P complex weights, staged in LDS
struct LwsTables {
  const float2* W;
  int Q, L, P, nfft, bins;
};
constexpr int kMaxWeights = 1024;     // complex entries of the LDS copy

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// one-sided row -> bin m in (-bins, nfft): conjugate symmetry of a real signal's spectrum
__device__ __forceinline__ float2 mirrored(const float2* row, int m, int bins, int nfft) {
  if (m < 0) { const float2 v = row[-m]; return make_float2(v.x, -v.y); }
  if (m >= bins) { const float2 v = row[nfft - m]; return make_float2(v.x, -v.y); }
  return row[m];
}

// mag * z / |z|, or `fallback` when z == 0
__device__ __forceinline__ float2 with_phase_of(float mag, float2 z, float2 fallback) {
  const float a = sqrtf(z.x * z.x + z.y * z.y);
  if (!(a > 0.f)) return fallback;
  const float s = mag / a;
  return make_float2(z.x * s, z.y * s);
}

constexpr int kRing = 8;       // frames in the LDS ring (>= 2Q - 1 = 7 for Q = 4)
constexpr int kMaxNf = 8;      // no-future threshold steps

struct CausalParams {
  float2* spec;          // [clips][T][bins], written (read first when use_init)
  const float* mag;      // [clips][T][bins]
  const float* mean_mag; // [clips]
  int T;
  int look_ahead;
  int nf_steps;
  float nf_thr[kMaxNf];  // multiples of the clip's mean magnitude, last one 0 (every bin set)
  int online_iterations;
  float on_alpha, on_beta;
  int use_init;          // != 0: spec holds initial phases (complex input of run_lws): frames are not re-initialised
};

// LDS: ring[kRing][bins] complex + the weight table.  QC / LC / PC > 0: compile-time Q, L, P (the reference's geometry
// 4, 5, 4: fully unrolled tap loops, mask instead of modulo); 0: run-time values.  P is a power of two (checked by the
// launcher), so m & (P - 1) is m mod P for negative m too.
template <int QC, int LC, int PC>
__global__ __launch_bounds__(576) void lws_causal_kernel(const CausalParams c, const LwsTables tb) {
  extern __shared__ __attribute__((aligned(16))) float2 lws_smem[];
  const int bins = tb.bins, nfft = tb.nfft, Q = QC ? QC : tb.Q, L = LC ? LC : tb.L;
  float2* ring = lws_smem;                    // [kRing][bins]
  float2* Ws = lws_smem + kRing * bins;       // [2Q - 1][2L - 1][P]
  const int P = PC ? PC : tb.P, nW = (2 * Q - 1) * (2 * L - 1) * P;
  for (int i = threadIdx.x; i < nW; i += blockDim.x) Ws[i] = tb.W[i];
  const int clip = blockIdx.x;
  const int f = threadIdx.x;
  const bool live = f < bins;
  float2* spec = c.spec + (int64_t)clip * c.T * bins;
  const float* mag = c.mag + (int64_t)clip * c.T * bins;
  const float ref = c.mean_mag[clip];
  const int KW = 2 * L - 1;

  for (int i = threadIdx.x; i < kRing * bins; i += blockDim.x) ring[i] = make_float2(0.f, 0.f);
  __syncthreads();
  int last_init = -1;          // newest frame whose ring slot holds valid data

  // sum over frames t + q, q in [q_lo, q_hi], of the truncated projection at bin f of frame t
  auto local_sum = [&](int t, int q_lo, int q_hi) -> float2 {
    float2 z = make_float2(0.f, 0.f);
#pragma unroll
    for (int q = -(Q - 1); q <= Q - 1; ++q) {
      const int tq = t + q;
      if (q < q_lo || q > q_hi || tq < 0 || tq >= c.T || tq > last_init) continue;
      const float2* row = ring + (tq & (kRing - 1)) * bins;
      const float2* Wq = Ws + (q + Q - 1) * KW * P;
#pragma unroll
      for (int p = -(L - 1); p < L; ++p) {
        if (q == 0 && p == 0) continue;
        const int m = f + p;
        const float2 x = mirrored(row, m, bins, nfft);
        const float2 wx = cmul(Wq[(p + L - 1) * P + (m & (P - 1))], x);
        z.x += wx.x; z.y += wx.y;
      }
    }
    return z;
  };

  // "no future": frame t from the frames before it (and its own bins set by earlier, higher thresholds)
  auto init_frame = [&](int t) {
    float2* slot = ring + (t & (kRing - 1)) * bins;
    if (c.use_init) {
      if (live) slot[f] = spec[(int64_t)t * bins + f];
      last_init = t;
      __syncthreads();
      return;
    }
    if (live) slot[f] = make_float2(0.f, 0.f);
    last_init = t;
    __syncthreads();
    for (int s = 0; s < c.nf_steps; ++s) {
      float2 nv = make_float2(0.f, 0.f);
      bool set = false;
      if (live) {
        const float a = mag[(int64_t)t * bins + f];
        if (a > c.nf_thr[s] * ref || c.nf_thr[s] <= 0.f) {
          nv = with_phase_of(a, local_sum(t, -(Q - 1), 0), make_float2(a, 0.f));     // zero phase when nothing is known yet
          set = true;
        }
      }
      __syncthreads();
      if (set) slot[f] = nv;
      __syncthreads();
    }
  };

  const int la = c.look_ahead;
  for (int t0 = 0; t0 <= la && t0 < c.T; ++t0) init_frame(t0);
  for (int t = 0; t < c.T; ++t) {
    const int ta = t + la;
    if (ta < c.T && ta > la) init_frame(ta);
    float2* slot = ring + (t & (kRing - 1)) * bins;
    for (int i = 0; i < c.online_iterations; ++i) {
      const float thr = c.online_iterations > 1 ? c.on_alpha * __expf(-c.on_beta * (float)i) * ref : 0.f;
      float2 nv = make_float2(0.f, 0.f);
      bool set = false;
      if (live) {
        const float a = mag[(int64_t)t * bins + f];
        if (a > thr) {
          nv = with_phase_of(a, local_sum(t, -(Q - 1), Q - 1), slot[f]);
          set = true;
        }
      }
      __syncthreads();
      if (set) slot[f] = nv;
      __syncthreads();
    }
    if (live) spec[(int64_t)t * bins + f] = slot[f];
  }
}

template <int QC, int LC, int PC>
__global__ __launch_bounds__(256) void lws_batch_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                        const float* __restrict__ mag, const float* __restrict__ mean_mag,
                                                        int T, float thr_mult, LwsTables tb, int64_t total) {
  __shared__ float2 Ws[kMaxWeights];
  const int bins = tb.bins, nfft = tb.nfft, Q = QC ? QC : tb.Q, L = LC ? LC : tb.L, KW = 2 * L - 1, P = PC ? PC : tb.P;
  for (int k = threadIdx.x; k < (2 * Q - 1) * KW * P; k += blockDim.x) Ws[k] = tb.W[k];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int f = (int)(i % bins);
  const int64_t row = i / bins;
  const int t = (int)(row % T);
  const int64_t clip = row / T;
  const float a = mag[i];
  const float2 cur = in[i];
  if (!(a > thr_mult * mean_mag[clip])) { out[i] = cur; return; }
  float2 z = make_float2(0.f, 0.f);
#pragma unroll
  for (int q = -(Q - 1); q < Q; ++q) {
    const int tq = t + q;
    if (tq < 0 || tq >= T) continue;
    const float2* src = in + (clip * T + tq) * bins;
    const float2* Wq = Ws + (q + Q - 1) * KW * P;
#pragma unroll
    for (int p = -(L - 1); p < L; ++p) {
      if (q == 0 && p == 0) continue;
      const int m = f + p;
      const float2 x = mirrored(src, m, bins, nfft);
      const float2 wx = cmul(Wq[(p + L - 1) * P + (m & (P - 1))], x);
      z.x += wx.x; z.y += wx.y;
    }
  }
  out[i] = with_phase_of(a, z, cur);
}


// ---------------------------------------------------------------------------------------------------------------------
// The reference's geometry (nfft 1024, hop 256: Q = 4, L = 5, P = 4, 513 bins) on kernels built for it (r3).  The r2
// kernels above spent their time in the tap loop: two LDS reads per tap (data + weight), a three-way branch for the
// conjugate-symmetric bins beyond the one-sided spectrum, and two workgroup barriers per Jacobi step -- ~10 us per step,
// 3 072 dependent steps per clip, one CU per clip: 35 of the 48 ms of a 64-clip batch.  Here
//   * a thread keeps the 63 complex weights of ITS bin in registers for the whole launch (the frame rotation depends on
//     (f + p) mod 4 only, i.e. on the thread);
//   * rows carry their L - 1 mirrored bins on both sides (written conjugated by the threads that own the source bins), so
//     a tap is one unconditional 8-byte LDS read at row[m + 4];
//   * the frame being refined is double-buffered (read one copy, write the other): one barrier per step;
//   * the 100 batch sweeps run on LDS tiles of 16 frames + 3 halo frames either side of one clip instead of 63 global
//     (L1 / L2) loads per bin.
// Both are bound by LDS bandwidth now (63 x 8 B per bin and step).
constexpr int kFH = 4;                     // mirrored bins either side of a row (L - 1)
constexpr int kFBins = 513;
constexpr int kFRow = kFBins + 2 * kFH;    // float2 per LDS row
constexpr int kFQ = 4, kFL = 5, kFKW = 9, kFP = 4;
#ifndef ADVOC_LWS_SWEEP_FRAMES
#define ADVOC_LWS_SWEEP_FRAMES 8
#endif
constexpr int kSweepFrames = ADVOC_LWS_SWEEP_FRAMES;            // frames per tile of the batch sweep
// advoc_lws_batch_sweeps_c64 documents tile_work as clips * ceil(nframes / 8) floats (include/advoc_hip.h; the Python
// binding allocates clips * ceil(nframes / 4)): a build with smaller tiles would overrun the caller's buffer
static_assert(kSweepFrames >= 8, "tile_work is sized for tiles of at least 8 frames");

// The weight of tap (q, p) at bin m = f + p is alpha_q(p) (-i)^(q m): the frame rotation exp(-2 pi i m q nhop / nfft) is a
// power of -i for hop = nfft / 4.  alpha_q(p) = W[q][p][0] is the same for every thread (scalar loads, no vector
// registers -- 63 complex weights per thread spilled 750 bytes of scratch per lane); the rotation is applied to sums:
//   sum_p alpha_q(p) (-i)^(q (f + p)) x[f + p] = (-i)^(q f) * sum_c (-i)^(q c) * [ sum_{p = c mod 4} alpha_q(p) x[f + p] ]
// -- compile-time quarter turns on the four class sums, one per-thread unit factor rot[q] = (-i)^(q f) per row.
typedef float f2v __attribute__((ext_vector_type(2)));
// c += al * x (complex) as TWO packed instructions: v_pk_fma_f32 takes the scalar weight, the half swap and the sign flip
// as operand modifiers
__device__ __forceinline__ void cmac(f2v& c, float2 al, f2v x) {
  c = __builtin_elementwise_fma((f2v){al.x, al.x}, x, c);
  c = __builtin_elementwise_fma((f2v){-al.y, al.y}, __builtin_shufflevector(x, x, 1, 0), c);
}
// t += (-i)^k v, k compile time
template <int K>
__device__ __forceinline__ void add_rot(f2v& t, f2v v) {
  if (K == 0) t += v;
  else if (K == 1) t += (f2v){v.y, -v.x};
  else if (K == 2) t -= v;
  else t += (f2v){-v.y, v.x};
}
constexpr int mod4(int v) { return ((v % 4) + 4) % 4; }

template <int QI, bool SKIP_CENTRE>      // QI = q + Q - 1
__device__ __forceinline__ void row_taps(const float2* __restrict__ row, int f, const float2* __restrict__ W, float2 rot,
                                         float2& z) {
  const f2v* r = reinterpret_cast<const f2v*>(row);
  f2v cs[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
  for (int p = -(kFL - 1); p <= kFL - 1; ++p) {
    if (SKIP_CENTRE && p == 0) continue;
    cmac(cs[p & 3], W[(QI * kFKW + p + kFL - 1) * kFP], r[f + p]);        // W: uniform address, scalar load
  }
  constexpr int q = QI - (kFQ - 1);
  f2v t = cs[0];
  add_rot<mod4(q * 1)>(t, cs[1]);
  add_rot<mod4(q * 2)>(t, cs[2]);
  add_rot<mod4(q * 3)>(t, cs[3]);
  f2v zz = {z.x, z.y};
  cmac(zz, rot, t);
  z = make_float2(zz.x, zz.y);
}

// rot[q + Q - 1] = (-i)^(q f)
__device__ __forceinline__ void bin_rotations(int f, float2 (&rot)[2 * kFQ - 1]) {
#pragma unroll
  for (int qi = 0; qi < 2 * kFQ - 1; ++qi) {
    const int k = ((qi - (kFQ - 1)) * (f & 3)) & 3;
    rot[qi] = make_float2(k == 0 ? 1.f : (k == 2 ? -1.f : 0.f), k == 1 ? -1.f : (k == 3 ? 1.f : 0.f));
  }
}

template <int QI>
__device__ __forceinline__ void row_dispatch(const float2* __restrict__ ring_row, const float2* __restrict__ cur, int f,
                                             const float2* __restrict__ W, const float2 (&rot)[2 * kFQ - 1], float2& z) {
  if (QI == kFQ - 1) row_taps<QI, true>(cur, f, W, rot[QI], z);
  else row_taps<QI, false>(ring_row, f, W, rot[QI], z);
}

// row[f] = v plus the mirrored copies that other bins read beyond the one-sided spectrum: bin m < 0 is conj(bin -m),
// bin m >= 513 is conj(bin 1024 - m)
__device__ __forceinline__ void store_bin(float2* __restrict__ row, int f, float2 v) {
  row[f] = v;
  const float2 cv = make_float2(v.x, -v.y);
  if (f >= 1 && f <= kFH) row[-f] = cv;
  if (f >= kFBins - 1 - kFH && f <= kFBins - 2) row[2 * (kFBins - 1) - f] = cv;
}

__global__ __launch_bounds__(576) void lws_causal_fast_kernel(const CausalParams c, const float2* __restrict__ W) {
  extern __shared__ __attribute__((aligned(16))) float2 lws_smem[];
  float2* ring = lws_smem + kFH;                       // [kRing] rows of kFRow, pointing at bin 0
  float2* alt = lws_smem + kRing * kFRow + kFH;        // second copy of the frame being worked on
  const int clip = blockIdx.x;
  const int f = threadIdx.x < kFBins ? threadIdx.x : kFBins - 1;     // (idle lanes shadow the last bin, never store)
  const bool live = threadIdx.x < kFBins;
  float2 rot[2 * kFQ - 1];
  bin_rotations(f, rot);
  float2* spec = c.spec + (int64_t)clip * c.T * kFBins;
  const float* mag = c.mag + (int64_t)clip * c.T * kFBins;
  const float ref = c.mean_mag[clip];
  for (int i = threadIdx.x; i < (kRing + 1) * kFRow; i += blockDim.x) lws_smem[i] = make_float2(0.f, 0.f);
  __syncthreads();
  int last_init = -1;

  // The OTHER frames' part of the sum for frame t: frames t + q, q in [q_lo, q_hi] \ {0}, that exist and hold data.  It
  // does not change while frame t is being refined (only t's own row does), so a refine() call takes it ONCE and every
  // Jacobi step adds the 8 taps of the frame's own row -- 54 + 8 n taps per frame instead of 63 n.
  auto others_sum = [&](int t, int q_lo, int q_hi) -> float2 {
    float2 z = make_float2(0.f, 0.f);
#define ADVOC_LWS_ROW(QI)                                                                                   \
    {                                                                                                       \
      const int q_ = (QI) - (kFQ - 1), tq_ = t + q_;                                                        \
      if (!(q_ < q_lo || q_ > q_hi || tq_ < 0 || tq_ >= c.T || tq_ > last_init))      /* uniform */          \
        row_dispatch<QI>(ring + (tq_ & (kRing - 1)) * kFRow, nullptr, f, W, rot, z);                        \
    }
    ADVOC_LWS_ROW(0) ADVOC_LWS_ROW(1) ADVOC_LWS_ROW(2) ADVOC_LWS_ROW(4) ADVOC_LWS_ROW(5) ADVOC_LWS_ROW(6)
#undef ADVOC_LWS_ROW
    return z;
  };

  // `steps` Jacobi steps on frame t, alternating between its ring row and `alt`; the result ends in the ring row
  auto refine = [&](int t, int steps, int q_lo, int q_hi, bool nofuture) {
    float2* home = ring + (t & (kRing - 1)) * kFRow;
    const float a = mag[(int64_t)t * kFBins + f];
    float2* src = home;
    float2* dst = alt;
    const float2 zo = steps > 0 ? others_sum(t, q_lo, q_hi) : make_float2(0.f, 0.f);
    for (int i = 0; i < steps; ++i) {
      float thr;
      if (nofuture) thr = c.nf_thr[i] > 0.f ? c.nf_thr[i] * ref : -1.f;
      else thr = steps > 1 ? c.on_alpha * __expf(-c.on_beta * (float)i) * ref : 0.f;
      const float2 old = src[f];
      float2 nv = old;
      if (a > thr) {
        float2 z = zo;
        row_dispatch<kFQ - 1>(src, src, f, W, rot, z);          // the frame's own row (q = 0 is inside every [q_lo, q_hi])
        nv = with_phase_of(a, z, nofuture ? make_float2(a, 0.f) : old);
      }
      if (live) store_bin(dst, f, nv);
      __syncthreads();
      float2* tmp = src; src = dst; dst = tmp;
    }
    if (src != home) {                 // odd number of steps: bring the result home
      if (live) store_bin(home, f, src[f]);
      __syncthreads();
    }
  };

  auto init_frame = [&](int t) {
    float2* home = ring + (t & (kRing - 1)) * kFRow;
    if (live) store_bin(home, f, c.use_init ? spec[(int64_t)t * kFBins + f] : make_float2(0.f, 0.f));
    last_init = t;
    __syncthreads();
    if (!c.use_init) refine(t, c.nf_steps, -(kFQ - 1), 0, true);
  };

  const int la = c.look_ahead;
  for (int t0 = 0; t0 <= la && t0 < c.T; ++t0) init_frame(t0);
  for (int t = 0; t < c.T; ++t) {
    const int ta = t + la;
    if (ta < c.T && ta > la) init_frame(ta);
    refine(t, c.online_iterations, -(kFQ - 1), kFQ - 1, false);
    if (live) spec[(int64_t)t * kFBins + f] = ring[(t & (kRing - 1)) * kFRow + f];
  }
}

// One batch sweep on LDS tiles: a workgroup = kSweepFrames frames of one clip (+ Q - 1 frames either side).
//
// A thread owns ONE bin and ALL the tile's frames: the value at (row R, bin f + p) is read from LDS once and feeds every
// frame tt = R - (Q - 1) - q it is a tap (q, p) of -- 14 x 9 = 126 reads for 8 x 63 = 504 complex MACs where one frame at
// a time re-read every value up to seven times (the sweeps were bound by exactly those reads).  For the weights to stay
// scalar the rotation R_q(f + p) must not depend on the lane: a WAVE holds bins of one residue class cl = f mod 4 (f = 4 j
// + cl, j = the lane's index), so W[q][p][(cl + p) mod 4] is one scalar load per (q, p), and rows are stored as four
// PLANES by bin mod 4 -- bin m at plane m & 3, index (m + 4) >> 2 -- so that the lanes of a wave read consecutive words
// (bins 4 apart in an interleaved row would hit 4 of the 32 banks).  Eight accumulators per thread, no per-thread weights.
constexpr int kPlane = 132;                     // float2 per plane: bins -4 .. 516 -> indices 0 .. 130
constexpr int kPRow = 4 * kPlane;               // float2 per planar row
__device__ __forceinline__ int planar(int m) { return (m & 3) * kPlane + ((m + 4) >> 2); }

__global__ __launch_bounds__(576) void lws_sweep_fast_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                             const float* __restrict__ mag,
                                                             const float* __restrict__ mean_mag, int T, float thr_mult,
                                                             const float2* __restrict__ W, int tiles_per_clip,
                                                             const float* __restrict__ tile_max) {
  extern __shared__ __attribute__((aligned(16))) float2 lws_smem[];
  constexpr int kRows = kSweepFrames + 2 * (kFQ - 1);
  float2* rows = lws_smem;
  const int clip = blockIdx.x / tiles_per_clip;
  // sparse mode (advoc_lws_batch_sweeps_c64): no bin of this tile is above the threshold -> nothing to do, `out` already
  // holds these bins (both buffers start equal and a bin below a non-increasing threshold has never been touched)
  if (tile_max && !(tile_max[blockIdx.x] > thr_mult * mean_mag[clip])) return;
  const int t0 = (blockIdx.x - clip * tiles_per_clip) * kSweepFrames;
  const float2* src = in + (int64_t)clip * T * kFBins;
  // ---- the tile's rows, coalesced (thread = bin), into the planar layout with the mirrored bins either side ----
  {
    const int b = threadIdx.x;
    if (b < kFBins) {
      float2 vr[kRows];                                    // every row's load in flight before the first LDS write
#pragma unroll
      for (int r = 0; r < kRows; ++r) {
        const int t = t0 - (kFQ - 1) + r;
        vr[r] = make_float2(0.f, 0.f);
        if (t >= 0 && t < T) vr[r] = src[(int64_t)t * kFBins + b];
      }
#pragma unroll
      for (int r = 0; r < kRows; ++r) {
        const float2 v = vr[r];
        float2* row = rows + r * kPRow;
        row[planar(b)] = v;
        const float2 cv = make_float2(v.x, -v.y);          // bin m < 0 is conj(bin -m), bin m >= 513 is conj(bin 1024 - m)
        if (b >= 1 && b <= kFH) row[planar(-b)] = cv;
        if (b >= kFBins - 1 - kFH && b <= kFBins - 2) row[planar(2 * (kFBins - 1) - b)] = cv;
      }
    }
  }
  // ---- wave -> residue class, lane -> index within it: waves 0-2 class 0 (129 bins), 3-4 / 5-6 / 7-8 classes 1 / 2 / 3 ----
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cl = wave < 3 ? 0 : 1 + ((wave - 3) >> 1);
  const int jn = cl == 0 ? 129 : 128;
  const int jraw = (wave < 3 ? wave : ((wave - 3) & 1)) * 64 + (threadIdx.x & 63);
  const bool live = jraw < jn;
  const int j = live ? jraw : jn - 1;                     // (idle lanes shadow the class's last bin, never store)
  const int f = 4 * j + cl;
  const float thr = thr_mult * mean_mag[clip];
  const int nt = T - t0 < kSweepFrames ? T - t0 : kSweepFrames;
  // the tile's magnitudes at this bin; a wave none of whose bins is above the threshold in any frame skips the taps
  float a[kSweepFrames];
  bool any = false;
#pragma unroll
  for (int tt = 0; tt < kSweepFrames; ++tt) {
    a[tt] = tt < nt ? mag[((int64_t)clip * T + t0 + tt) * kFBins + f] : 0.f;
    any = any || a[tt] > thr;
  }
  const bool wave_active = __ballot(any && live) != 0;
  __syncthreads();
  f2v z[kSweepFrames];
#pragma unroll
  for (int tt = 0; tt < kSweepFrames; ++tt) z[tt] = (f2v){0.f, 0.f};
#pragma unroll 1                                          // (unrolled, the scheduler hoists all 126 reads and spills)
  for (int p = wave_active ? -(kFL - 1) : kFL; p <= kFL - 1; ++p) {
    const int cp = cl + p;                                // uniform
    const int off = (cp & 3) * kPlane + ((cp + 4) >> 2) + j;
    float2 w[2 * kFQ - 1];
#pragma unroll
    for (int qi = 0; qi < 2 * kFQ - 1; ++qi) w[qi] = W[(qi * kFKW + p + kFL - 1) * kFP + (cp & 3)];      // scalar loads
    if (p == 0) w[kFQ - 1] = make_float2(0.f, 0.f);       // the bin itself is not a tap
#pragma unroll
    for (int R = 0; R < kRows; ++R) {
      const f2v x = *reinterpret_cast<const f2v*>(rows + R * kPRow + off);
#pragma unroll
      for (int qi = 0; qi < 2 * kFQ - 1; ++qi) {
        const int tt = R - qi;                            // row R = frame tt + (Q - 1) + q, q = qi - (Q - 1)
        if (tt < 0 || tt >= kSweepFrames) continue;
        cmac(z[tt], w[qi], x);
      }
    }
  }
#pragma unroll
  for (int tt = 0; tt < kSweepFrames; ++tt) {
    if (tt >= nt) break;
    const int64_t i = ((int64_t)clip * T + t0 + tt) * kFBins + f;
    float2 v = rows[(tt + kFQ - 1) * kPRow + planar(f)];
    if (a[tt] > thr) v = with_phase_of(a[tt], make_float2(z[tt].x, z[tt].y), v);
    if (live) out[i] = v;
  }
}

// tile_max[clip * tiles_per_clip + tile] = largest magnitude of the tile's frames
__global__ __launch_bounds__(256) void lws_tile_max_kernel(const float* __restrict__ mag, int T, int tiles_per_clip,
                                                           float* __restrict__ tile_max) {
  const int clip = blockIdx.x / tiles_per_clip;
  const int t0 = (blockIdx.x - clip * tiles_per_clip) * kSweepFrames;
  const int nt = T - t0 < kSweepFrames ? T - t0 : kSweepFrames;
  const float* src = mag + ((int64_t)clip * T + t0) * kFBins;
  float m = 0.f;
  for (int i = threadIdx.x; i < nt * kFBins; i += 256) m = fmaxf(m, src[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) tile_max[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// mean_mag[clip] = mean of mag[clip][:][:] (double accumulation; 1024 threads per clip, 16-byte loads where the clip's
// slice allows: 256 threads with scalar loads took 176 us per call, 2 % of a 64-clip LWS batch)
__global__ __launch_bounds__(1024) void lws_mean_kernel(const float* __restrict__ mag, int64_t per_clip, float* __restrict__ mean_mag) {
  const float* src = mag + (int64_t)blockIdx.x * per_clip;
  double s = 0.0;
  if (per_clip % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    const float4* src4 = reinterpret_cast<const float4*>(src);
    for (int64_t i = threadIdx.x; i < per_clip / 4; i += blockDim.x) {
      const float4 v = src4[i];
      s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
    }
  } else {
    for (int64_t i = threadIdx.x; i < per_clip; i += blockDim.x) s += (double)src[i];
  }
  __shared__ double red[1024];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) mean_mag[blockIdx.x] = (float)(red[0] / (double)per_clip);
}

}  // namespace

extern "C" int advoc_lws_mean_mag_f32(const float* mag, int64_t clips, int64_t per_clip, float* mean_mag,
                                      advoc_stream_t stream) {
  if (clips < 0 || per_clip <= 0) return ADVOC_ERR_BAD_SHAPE;
  if (clips == 0) return ADVOC_OK;
  if (!mag || !mean_mag) return ADVOC_ERR_NULL;
  if (clips > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(lws_mean_kernel, dim3((unsigned)clips), dim3(1024), 0, advoc::as_stream(stream), mag, per_clip, mean_mag);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

namespace {
// ADVOC_LWS_GENERIC=1: every geometry on the kernels that read the caller's whole weight table (see advoc_lws_causal_c64)
bool lws_generic_only() {
  const char* e = getenv("ADVOC_LWS_GENERIC");      // (read per call: three C calls per vocoded batch)
  return e && atoi(e) != 0;
}
}  // namespace

extern "C" int advoc_lws_causal_c64(float* spec, const float* mag, const float* mean_mag, int64_t clips, int64_t nframes,
                                    int32_t nfft, int32_t nhop, const float* weights, int32_t period, int32_t L,
                                    int32_t look_ahead, const float* nofuture_thresholds_host, int32_t nofuture_steps,
                                    int32_t online_iterations, float online_alpha, float online_beta, int32_t use_init,
                                    advoc_stream_t stream) {
  if (clips < 0 || nframes < 0 || nfft < 4 || nhop < 1 || nhop > nfft) return ADVOC_ERR_BAD_SHAPE;
  if (clips == 0 || nframes == 0) return ADVOC_OK;
  if (!spec || !mag || !mean_mag || !weights || (nofuture_steps > 0 && !nofuture_thresholds_host)) return ADVOC_ERR_NULL;
  const int bins = nfft / 2 + 1;
  const int Q = (nfft + nhop - 1) / nhop;
  if (period < 1 || (period & (period - 1)) || (int64_t)nhop * period % nfft || (2 * Q - 1) * (2 * L - 1) * period > kMaxWeights) return ADVOC_ERR_UNSUPPORTED;
  if (nfft % 2 || bins > 576 || 2 * Q - 1 > kRing || L < 1 || L > 16 || look_ahead < 0 || look_ahead > Q - 1 ||
      nofuture_steps < 1 || nofuture_steps > kMaxNf || online_iterations < 0 || clips > 0x7fffffffLL ||
      nframes > 0x7fffffffLL)
    return ADVOC_ERR_UNSUPPORTED;
  CausalParams c = {};
  c.spec = reinterpret_cast<float2*>(spec); c.mag = mag; c.mean_mag = mean_mag;
  c.T = (int)nframes; c.look_ahead = look_ahead; c.nf_steps = nofuture_steps;
  for (int i = 0; i < nofuture_steps; ++i) c.nf_thr[i] = nofuture_thresholds_host[i];
  c.online_iterations = online_iterations; c.on_alpha = online_alpha; c.on_beta = online_beta; c.use_init = use_init;
  LwsTables tb = {reinterpret_cast<const float2*>(weights), Q, L, period, nfft, bins};
  const size_t lds = sizeof(float2) * ((size_t)kRing * bins + (size_t)(2 * Q - 1) * (2 * L - 1) * period);
  ADVOC_CLEAR_LAUNCH_ERROR();
  // The kernels for the reference geometry read ONLY weights[q][p][0] and apply the frame rotation (-i)^(q r) themselves
  // (include/advoc_hip.h states the table form they rely on); ADVOC_LWS_GENERIC=1 sends every geometry through the
  // kernels that read the whole table -- for callers with a table of another form
  if (!lws_generic_only() && Q == kFQ && L == kFL && period == kFP && bins == kFBins && nofuture_steps <= kMaxNf) {
    constexpr size_t lds_fast = sizeof(float2) * (size_t)(kRing + 1) * kFRow;
    hipLaunchKernelGGL(lws_causal_fast_kernel, dim3((unsigned)clips), dim3(576), lds_fast, advoc::as_stream(stream), c,
                       reinterpret_cast<const float2*>(weights));
  } else if (Q == 4 && L == 5 && period == 4)
    hipLaunchKernelGGL((lws_causal_kernel<4, 5, 4>), dim3((unsigned)clips), dim3(576), lds, advoc::as_stream(stream), c, tb);
  else
    hipLaunchKernelGGL((lws_causal_kernel<0, 0, 0>), dim3((unsigned)clips), dim3(576), lds, advoc::as_stream(stream), c, tb);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

extern "C" int advoc_lws_batch_c64(const float* spec_in, float* spec_out, const float* mag, const float* mean_mag,
                                   int64_t clips, int64_t nframes, int32_t nfft, int32_t nhop, const float* weights,
                                   int32_t period, int32_t L, float threshold, advoc_stream_t stream) {
  if (clips < 0 || nframes < 0 || nfft < 4 || nhop < 1 || nhop > nfft) return ADVOC_ERR_BAD_SHAPE;
  if (clips == 0 || nframes == 0) return ADVOC_OK;
  if (!spec_in || !spec_out || !mag || !mean_mag || !weights) return ADVOC_ERR_NULL;
  if (spec_in == spec_out) return ADVOC_ERR_UNSUPPORTED;      // a sweep reads the previous iterate of every neighbour
  const int bins = nfft / 2 + 1;
  const int Q = (nfft + nhop - 1) / nhop;
  if (nfft % 2 || L < 1 || L > 16 || nframes > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
  if (period < 1 || (period & (period - 1)) || (int64_t)nhop * period % nfft || (2 * Q - 1) * (2 * L - 1) * period > kMaxWeights) return ADVOC_ERR_UNSUPPORTED;
  const int64_t total = clips * nframes * bins;
  const int64_t blocks = advoc::ceil_div(total, 256);
  if (blocks > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
  LwsTables tb = {reinterpret_cast<const float2*>(weights), Q, L, period, nfft, bins};
  ADVOC_CLEAR_LAUNCH_ERROR();
  if (!lws_generic_only() && Q == kFQ && L == kFL && period == kFP && bins == kFBins) {
    const int64_t tiles_per_clip = advoc::ceil_div(nframes, kSweepFrames);
    if (clips * tiles_per_clip > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
    constexpr size_t lds_sweep = sizeof(float2) * (size_t)(kSweepFrames + 2 * (kFQ - 1)) * kPRow;
    // (per call, not cached in a static: the attribute is per device)
    const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(lws_sweep_fast_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sweep);
    if (attr != hipSuccess) { advoc::note_hip_error(attr); return ADVOC_ERR_HIP; }
    hipLaunchKernelGGL(lws_sweep_fast_kernel, dim3((unsigned)(clips * tiles_per_clip)), dim3(576), lds_sweep,
                       advoc::as_stream(stream), reinterpret_cast<const float2*>(spec_in),
                       reinterpret_cast<float2*>(spec_out), mag, mean_mag, (int)nframes, threshold,
                       reinterpret_cast<const float2*>(weights), (int)tiles_per_clip, (const float*)nullptr);
  } else if (Q == 4 && L == 5 && period == 4)
    hipLaunchKernelGGL((lws_batch_kernel<4, 5, 4>), dim3((unsigned)blocks), dim3(256), 0, advoc::as_stream(stream),
                       reinterpret_cast<const float2*>(spec_in), reinterpret_cast<float2*>(spec_out), mag, mean_mag,
                       (int)nframes, threshold, tb, total);
  else
    hipLaunchKernelGGL((lws_batch_kernel<0, 0, 0>), dim3((unsigned)blocks), dim3(256), 0, advoc::as_stream(stream),
                       reinterpret_cast<const float2*>(spec_in), reinterpret_cast<float2*>(spec_out), mag, mean_mag,
                       (int)nframes, threshold, tb, total);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

// All batch sweeps of run_lws in one call: spec_a holds the result of the time-ordered pass and receives the final
// spectrogram, spec_b is a work buffer of the same size, tile_work >= clips * ceil(nframes / 8) floats.  With
// non-increasing thresholds (the reference's schedule alpha * exp(-beta * i^gamma)) on the reference geometry the sweeps
// are SPARSE: both buffers start equal, a tile none of whose bins exceeds the sweep's threshold is skipped outright (its
// bins have never changed), so the early sweeps -- thresholds far above most of a speech spectrogram -- cost what their
// active tiles cost instead of a full pass over memory.  Otherwise: one dense advoc_lws_batch_c64 sweep per threshold.
extern "C" int advoc_lws_batch_sweeps_c64(float* spec_a, float* spec_b, const float* mag, const float* mean_mag,
                                          int64_t clips, int64_t nframes, int32_t nfft, int32_t nhop, const float* weights,
                                          int32_t period, int32_t L, const float* thresholds_host, int32_t n_sweeps,
                                          float* tile_work, advoc_stream_t stream) {
  if (clips < 0 || nframes < 0 || n_sweeps < 0 || nfft < 4 || nhop < 1 || nhop > nfft) return ADVOC_ERR_BAD_SHAPE;
  if (clips == 0 || nframes == 0 || n_sweeps == 0) return ADVOC_OK;
  if (!spec_a || !spec_b || !mag || !mean_mag || !weights || !thresholds_host) return ADVOC_ERR_NULL;
  if (spec_a == spec_b) return ADVOC_ERR_UNSUPPORTED;
  const int bins = nfft / 2 + 1;
  const int Q = (nfft + nhop - 1) / nhop;
  hipStream_t st = advoc::as_stream(stream);
  bool sparse = !lws_generic_only() && Q == kFQ && L == kFL && period == kFP && bins == kFBins && tile_work != nullptr;
  for (int i = 1; i < n_sweeps && sparse; ++i) sparse = thresholds_host[i] <= thresholds_host[i - 1];
  const size_t bytes = sizeof(float2) * (size_t)clips * (size_t)nframes * (size_t)bins;
  float* cur = spec_a;
  float* nxt = spec_b;
  if (!sparse) {
    for (int i = 0; i < n_sweeps; ++i) {
      const int rc = advoc_lws_batch_c64(cur, nxt, mag, mean_mag, clips, nframes, nfft, nhop, weights, period, L,
                                         thresholds_host[i], stream);
      if (rc != ADVOC_OK) return rc;
      float* t = cur; cur = nxt; nxt = t;
    }
  } else {
    if (nframes > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
    const int64_t tiles_per_clip = advoc::ceil_div(nframes, kSweepFrames);
    if (clips * tiles_per_clip > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
    hipError_t e = hipMemcpyAsync(spec_b, spec_a, bytes, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) { advoc::note_hip_error(e); return ADVOC_ERR_HIP; }
    ADVOC_CLEAR_LAUNCH_ERROR();
    hipLaunchKernelGGL(lws_tile_max_kernel, dim3((unsigned)(clips * tiles_per_clip)), dim3(256), 0, st, mag, (int)nframes,
                       (int)tiles_per_clip, tile_work);
    ADVOC_RETURN_IF_LAUNCH_FAILED();
    constexpr size_t lds_sweep = sizeof(float2) * (size_t)(kSweepFrames + 2 * (kFQ - 1)) * kPRow;
    const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(lws_sweep_fast_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sweep);
    if (attr != hipSuccess) { advoc::note_hip_error(attr); return ADVOC_ERR_HIP; }
    for (int i = 0; i < n_sweeps; ++i) {
      ADVOC_CLEAR_LAUNCH_ERROR();
      hipLaunchKernelGGL(lws_sweep_fast_kernel, dim3((unsigned)(clips * tiles_per_clip)), dim3(576), lds_sweep, st,
                         reinterpret_cast<const float2*>(cur), reinterpret_cast<float2*>(nxt), mag, mean_mag, (int)nframes,
                         thresholds_host[i], reinterpret_cast<const float2*>(weights), (int)tiles_per_clip,
                         (const float*)tile_work);
      ADVOC_RETURN_IF_LAUNCH_FAILED();
      float* t = cur; cur = nxt; nxt = t;
    }
  }
  if (cur != spec_a) {          // odd number of sweeps: the newest iterate is in spec_b
    hipError_t e = hipMemcpyAsync(spec_a, cur, bytes, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) { advoc::note_hip_error(e); return ADVOC_ERR_HIP; }
  }
  return ADVOC_OK;
}
