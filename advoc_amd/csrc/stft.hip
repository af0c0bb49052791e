// STFT-magnitude feature extractor for gfx950 (MI355X).
//
// Replaces tf.contrib.signal.stft (+ tf.abs) reached from the reference at
// advoc/spectral.py:60-83 and advoc/loader.py:116-128: frame 1024 / hop 256, multiply by the
// lws sqrt-Hann window, real FFT, |.|.
//
// Mapping to the hardware
//   * one wavefront = TWO consecutive frames of one clip at a time, carried in the two halves of
//     2-wide fp32 vectors so that every butterfly / twiddle multiply is one packed instruction
//     (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) for both frames: the kernel is bound by VALU
//     issue (a 1024-point FFT per 3 KB of traffic), not by HBM, so halving the instruction count
//     per frame is what moves it up the bandwidth roofline.
//   * The 1024-point real FFT is a 512-point complex FFT (z[n] = x[2n] + i x[2n+1]) held 8 complex
//     values per lane, computed as three radix-8 passes (512 = 8*8*8); the two inter-pass
//     transposes go through a private, bank-conflict-free LDS plane per wave (no workgroup
//     barrier anywhere).
//   * samples are read straight from global memory, 8 bytes per lane on 512-byte rows; the 75 %
//     overlap between neighbouring frames is served by L1/L2.
//   * the window and the pass twiddles (they depend only on the lane) sit in LDS, the split twiddles in
//     registers; one transpose plane per wave serves the real and then the imaginary parts: 126 VGPRs and
//     30 KB of LDS per workgroup, i.e. four waves per SIMD.  A wave grid-strides over many frame pairs.
//   * the real-FFT split pairs Z[k] with Z[512-k] by one cross-lane permute per value, then
//     each lane stores 8 magnitudes per frame: 256 B contiguous per store instruction.
// Algorithmic traffic is 256 new samples in + 513 floats out per frame.
#include <math.h>

#include <stdlib.h>

#include "common.h"
#include "lds_dma.h"

namespace {

using advoc::wave_lds_sync;

constexpr int kNfft = 1024;
constexpr int kBins = kNfft / 2 + 1;
constexpr int kWaves = 4;
constexpr int kPlane = 576;  // 64 rows x 9 elements (8 + 1 pad): conflict-free transposes

typedef float f2 __attribute__((ext_vector_type(2)));   // .x = first frame of the pair, .y = second

// forward 8-point DFT in place: v[p] = sum_a v[a] * exp(-2*pi*i*a*p/8)
template <typename T>
__device__ __forceinline__ void dft8(T (&re)[8], T (&im)[8]) {
  const float h = 0.70710678118654752440f;
  // radix-2 DIF stage: sums feed even outputs, twiddled differences feed odd outputs
  T sr[4], si[4], dr[4], di[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    sr[n] = re[n] + re[n + 4];
    si[n] = im[n] + im[n + 4];
    dr[n] = re[n] - re[n + 4];
    di[n] = im[n] - im[n + 4];
  }
  // d[n] *= W8^n
  {
    T r1 = (dr[1] + di[1]) * h, i1 = (di[1] - dr[1]) * h;  // * (1 - i)/sqrt2
    dr[1] = r1; di[1] = i1;
    T r2 = di[2], i2 = -dr[2];                             // * (-i)
    dr[2] = r2; di[2] = i2;
    T r3 = (di[3] - dr[3]) * h, i3 = -(dr[3] + di[3]) * h; // * (-1 - i)/sqrt2
    dr[3] = r3; di[3] = i3;
  }
  // 4-point DFT of s -> even bins, of d -> odd bins
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    T* xr = half ? dr : sr;
    T* xi = half ? di : si;
    T b0r = xr[0] + xr[2], b0i = xi[0] + xi[2];
    T b2r = xr[0] - xr[2], b2i = xi[0] - xi[2];
    T b1r = xr[1] + xr[3], b1i = xi[1] + xi[3];
    T b3r = xi[1] - xi[3], b3i = -(xr[1] - xr[3]);  // (x1 - x3) * (-i)
    re[0 + half] = b0r + b1r; im[0 + half] = b0i + b1i;
    re[4 + half] = b0r - b1r; im[4 + half] = b0i - b1i;
    re[2 + half] = b2r + b3r; im[2 + half] = b2i + b3i;
    re[6 + half] = b2r - b3r; im[6 + half] = b2i - b3i;
  }
}

template <bool kComplexOut>
__global__ __launch_bounds__(kWaves * 64) void stft1024_kernel(
    const float* __restrict__ wav, int64_t nsamps, const float* __restrict__ window,
    const float2* __restrict__ twiddle, int nhop, int64_t nframes, float* __restrict__ out,
    int pairs_per_clip, int64_t total_pairs) {
  // ONE transpose plane per wave, used for the real and then the imaginary parts: half the LDS, so that
  // four workgroups (instead of three) fit a CU
  __shared__ f2 planes[kWaves][kPlane];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  f2* plane = &planes[wave][0];

  // ---- per-lane constants ----
  // They depend on the lane only.  The split twiddles stay in registers; the window and the
  // pass-1 / pass-2 twiddles sit in LDS ([j][lane], conflict-free 8-byte reads) -- holding all 64
  // of them in registers next to two frames of data left room for only 2 waves per SIMD.
  __shared__ float2 s_win[8][64], s_t1[8][64], s_t2[8][64];
  const int hi = lane >> 3, lo = lane & 7;
  // twiddles: cos/sin(2 pi e / 1024) from the 1024-entry table built once on the host in double
  // precision (float2 {cos, sin} per entry, L2 resident): W_N^e = tw[e * 1024 / N] conjugated
  float tsn[8], tcs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (wave == 0) {
      s_win[j][lane] = *reinterpret_cast<const float2*>(window + 128 * j + 2 * lane);
      // pass-1 twiddle W64^(b*p): lane = (b=hi, c=lo), p = j
      const float2 a = twiddle[((hi * j) & 63) * 16];
      s_t1[j][lane] = make_float2(a.x, -a.y);
      // pass-2 twiddle W512^(c*(p+8q)): lane = (p=hi, c=lo), q = j
      const float2 b = twiddle[((lo * (hi + 8 * j)) & 511) * 2];
      s_t2[j][lane] = make_float2(b.x, -b.y);
    }
    // split twiddle: theta = 2*pi*k/1024, k = lane + 64 j  (k <= 511)
    const float2 c = twiddle[lane + 64 * j];
    tcs[j] = c.x; tsn[j] = c.y;
  }
  __syncthreads();
  const int partner = (64 - lane) & 63;
  const bool aligned = ((nsamps | nhop) & 1) == 0;     // float2 loads stay 8-byte aligned

  // raw samples of a frame pair: x[2n], x[2n+1] for n = 64 a + lane, zero beyond the end of the clip
  // (pad_end).  Fetched one pair AHEAD: a wave has nothing else to overlap its own load latency with.
  float2 raw0[8], raw1[8];
#define ADVOC_STFT_FETCH(PAIR)                                                                        \
  {                                                                                                   \
    const int64_t clip_ = (PAIR) / pairs_per_clip;                                                    \
    const int64_t f_ = ((PAIR) - clip_ * pairs_per_clip) * 2;                                         \
    const bool two_ = f_ + 1 < nframes;                                                               \
    const float* src_ = wav + clip_ * nsamps;                                                         \
    const int64_t s0_ = f_ * nhop;                                                                    \
    if (aligned && s0_ + nhop + kNfft <= nsamps) {                                                    \
      _Pragma("unroll") for (int a = 0; a < 8; ++a) {                                                 \
        raw0[a] = *reinterpret_cast<const float2*>(src_ + s0_ + 128 * a + 2 * lane);                  \
        raw1[a] = *reinterpret_cast<const float2*>(src_ + s0_ + nhop + 128 * a + 2 * lane);           \
      }                                                                                               \
    } else {                                                                                          \
      _Pragma("unroll") for (int a = 0; a < 8; ++a) {                                                 \
        const int64_t i0 = s0_ + 128 * a + 2 * lane, i1 = i0 + nhop;                                  \
        raw0[a].x = i0 < nsamps ? src_[i0] : 0.f;                                                     \
        raw0[a].y = i0 + 1 < nsamps ? src_[i0 + 1] : 0.f;                                             \
        raw1[a].x = (two_ && i1 < nsamps) ? src_[i1] : 0.f;                                           \
        raw1[a].y = (two_ && i1 + 1 < nsamps) ? src_[i1 + 1] : 0.f;                                   \
      }                                                                                               \
    }                                                                                                 \
  }
  const int64_t pair0 = (int64_t)blockIdx.x * kWaves + wave;
  const int64_t pstride = (int64_t)gridDim.x * kWaves;
  for (int64_t pair = pair0; pair < total_pairs; pair += pstride) {
    const int64_t clip = pair / pairs_per_clip;
    const int64_t f = (pair - clip * pairs_per_clip) * 2;     // frames f and f + 1
    const bool two = f + 1 < nframes;

    ADVOC_STFT_FETCH(pair);
    f2 re[8], im[8];
    // z[n] = x[2n] w[2n] + i x[2n+1] w[2n+1]
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const float2 w = s_win[a][lane];
      re[a] = f2{raw0[a].x, raw1[a].x} * w.x;
      im[a] = f2{raw0[a].y, raw1[a].y} * w.y;
    }

    // pass 1: DFT over a -> p, twiddle, transpose (b,c | p) -> (p,c | b)
    dft8(re, im);
    {
      f2 ti[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const float2 t = s_t1[p][lane];
        const f2 r = re[p] * t.x - im[p] * t.y;
        ti[p] = re[p] * t.y + im[p] * t.x;
        plane[(8 * p + hi) * 9 + lo] = r;
      }
      wave_lds_sync();
#pragma unroll
      for (int b = 0; b < 8; ++b) re[b] = plane[(8 * hi + b) * 9 + lo];
      wave_lds_sync();
#pragma unroll
      for (int p = 0; p < 8; ++p) plane[(8 * p + hi) * 9 + lo] = ti[p];
      wave_lds_sync();
#pragma unroll
      for (int b = 0; b < 8; ++b) im[b] = plane[(8 * hi + b) * 9 + lo];
      wave_lds_sync();
    }

    // pass 2: DFT over b -> q, twiddle, transpose (p,c | q) -> (q,p | c)
    dft8(re, im);
    {
      f2 ti[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float2 t = s_t2[q][lane];
        const f2 r = re[q] * t.x - im[q] * t.y;
        ti[q] = re[q] * t.y + im[q] * t.x;
        plane[(8 * q + hi) * 9 + lo] = r;
      }
      wave_lds_sync();
#pragma unroll
      for (int c = 0; c < 8; ++c) re[c] = plane[lane * 9 + c];
      wave_lds_sync();
#pragma unroll
      for (int q = 0; q < 8; ++q) plane[(8 * q + hi) * 9 + lo] = ti[q];
      wave_lds_sync();
#pragma unroll
      for (int c = 0; c < 8; ++c) im[c] = plane[lane * 9 + c];
      wave_lds_sync();
    }

    // pass 3: DFT over c -> r.  Lane now holds Z[lane + 64 r].
    dft8(re, im);

    // real-FFT split: X[k] = ((Zk + conj(Zm)) - i W1024^k (Zk - conj(Zm))) / 2, m = 512 - k
    float* orow0 = out + ((clip * nframes + f) * kBins) * (kComplexOut ? 2 : 1);
    float* orow1 = orow0 + kBins * (kComplexOut ? 2 : 1);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      f2 c, d;
      c.x = __shfl(re[7 - r].x, partner, 64); c.y = __shfl(re[7 - r].y, partner, 64);
      d.x = __shfl(im[7 - r].x, partner, 64); d.y = __shfl(im[7 - r].y, partner, 64);
      if (lane == 0) {  // k = 64 r pairs with 512 - 64 r = 64 (8 - r) on the same lane
        c = re[(8 - r) & 7];
        d = im[(8 - r) & 7];
      }
      const f2 a = re[r], b = im[r];
      const f2 sr = a + c, si = b - d, dr = a - c, di = b + d;
      const f2 xr = 0.5f * (sr - (tsn[r] * dr - tcs[r] * di));
      const f2 xi = 0.5f * (si - (tsn[r] * di + tcs[r] * dr));
      const int k = lane + 64 * r;
      if (kComplexOut) {
        *reinterpret_cast<float2*>(orow0 + 2 * k) = make_float2(xr.x, xi.x);
        if (two) *reinterpret_cast<float2*>(orow1 + 2 * k) = make_float2(xr.y, xi.y);
      } else {
        const f2 m2 = xr * xr + xi * xi;
        orow0[k] = __builtin_amdgcn_sqrtf(m2.x);   // v_sqrt_f32, <= 1 ulp
        if (two) orow1[k] = __builtin_amdgcn_sqrtf(m2.y);
      }
    }
    if (lane == 0) {  // Nyquist bin: X[512] = Re Z0 - Im Z0
      const f2 xn = re[0] - im[0];
      if (kComplexOut) {
        *reinterpret_cast<float2*>(orow0 + 2 * (kBins - 1)) = make_float2(xn.x, 0.f);
        if (two) *reinterpret_cast<float2*>(orow1 + 2 * (kBins - 1)) = make_float2(xn.y, 0.f);
      } else {
        orow0[kBins - 1] = fabsf(xn.x);
        if (two) orow1[kBins - 1] = fabsf(xn.y);
      }
    }
  }
#undef ADVOC_STFT_FETCH
}

// ---------------------------------------------------------------------------------------------
// (r4) The same transform for hop = 256 = nfft / 4 -- the geometry of every preset the train step and the vocoder use --
// with fewer instructions per frame pair.  stft1024_kernel above is bound by what its waves ISSUE (rocprofv3:
// SQ_ACTIVE_INST_ANY x 4 waves per SIMD = 0.95; ~460 vector + 76 LDS instructions per pair, the LDS pipe 69 % busy with
// them), not by HBM, so the only way up the bandwidth roofline is to issue less:
//   * frames f and f + 1 overlap in 768 of their 1 024 samples: ten 8-byte loads per lane and pair instead of sixteen;
//   * the real-FFT split pairs bin k with bin 512 - k, and X[512 - k] = conj((S + T)) where X[k] = S - T: a lane computes
//     BOTH bins of the pairs whose lower bin it owns (k = lane + 64 r, r < 4), so S, T are formed once per pair of bins and
//     only half of the partner values cross lanes (16 ds_bpermute instead of 32); the upper bins are stored in descending
//     lane order (still one contiguous 252-byte run per instruction); lane 0 adds bin 256, which pairs with itself;
//   * the split's factor 1/2 is folded into the window table (a power of two: the same bits come out);
//   * the transposes read with ds_read_b64, one instruction per value: the ds_read2_b64 the compiler fuses two reads into
//     costs 8 LDS cycles per 1 KB against 2 per 512 B (MI355X_MICROARCH.md, LDS table), so the reads are inline assembly the
//     load/store optimiser does not see, tied to ONE hand-placed s_waitcnt;
//   * which pair a wave works on is wave-uniform: (clip, pair of the clip) live in scalar registers and advance by a fixed
//     step (no 64-bit division, no 64-bit vector address arithmetic per load and store -- ~80 vector instructions per pair
//     in the kernel above).
// Measured (r4, tools/micro/stft_variants.py, profiles/r04_b_stft_variants.txt): 132.9 -> 109.6 us for 512 clips (0.38 -> 0.46
// of 8 TB/s), 37.7 -> 30.0 us at the training feed (0.34 -> 0.42).  tools/micro/valu_rate.hip gives the price list the
// count is weighed with (ns per wave-instruction at a SIMD, four waves per SIMD): v_mov / v_add / v_fma_f32 1.2-1.3, packed
// f32 2.0-2.1 (a packed operation is 1.6 plain ones, not 1), v_sqrt_f32 3.5, v_cndmask_b32 on an SGPR mask 2.1 but 9.5 (!) in
// its VCC form, ds_read_b64 3.9, ds_write_b64 10.7, ds_bpermute_b32 10.1: per pair ~0.9 us of vector and ~0.8 us of LDS work
// in 1.7 us -- the two do not overlap, and starting the waves of a CU out of phase only adds the delay as a tail.
// kV bit 0 (measured slower, 115 us): real and imaginary parts through TWO planes per wave in one round -- half the waits,
// but 49 KB of LDS per workgroup, three workgroups per CU.  Variant 8 below (38 v_mov and 20 packed operations fewer):
// 110.8 us -- the count is no longer what binds.
// Same radix-8 x 3 arithmetic as above; results differ from stft1024_kernel in the last bit only (the split's association).
// ---------------------------------------------------------------------------------------------
template <int OFF>
__device__ __forceinline__ f2 lds_read_b64(unsigned addr) {
  f2 v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
#define ADVOC_LDS_READ8(DST, ADDR, BASE, STEP)                                                                     \
  {                                                                                                                \
    DST[0] = lds_read_b64<(BASE) + 0 * (STEP)>(ADDR); DST[1] = lds_read_b64<(BASE) + 1 * (STEP)>(ADDR);            \
    DST[2] = lds_read_b64<(BASE) + 2 * (STEP)>(ADDR); DST[3] = lds_read_b64<(BASE) + 3 * (STEP)>(ADDR);            \
    DST[4] = lds_read_b64<(BASE) + 4 * (STEP)>(ADDR); DST[5] = lds_read_b64<(BASE) + 5 * (STEP)>(ADDR);            \
    DST[6] = lds_read_b64<(BASE) + 6 * (STEP)>(ADDR); DST[7] = lds_read_b64<(BASE) + 7 * (STEP)>(ADDR);            \
  }
// every LDS read of this wave has returned; the eight values are "written" by the statement, so nothing that uses them can
// be scheduled above it
#define ADVOC_LDS_WAIT8(V)                                                                                          \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                               \
               : "+v"(V[0]), "+v"(V[1]), "+v"(V[2]), "+v"(V[3]), "+v"(V[4]), "+v"(V[5]), "+v"(V[6]), "+v"(V[7])     \
               :: "memory")
#define ADVOC_LDS_TIE8(V)                                                                                           \
  asm volatile("" : "+v"(V[0]), "+v"(V[1]), "+v"(V[2]), "+v"(V[3]), "+v"(V[4]), "+v"(V[5]), "+v"(V[6]), "+v"(V[7]))

__device__ __forceinline__ float bperm(int byte_index, float v) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_index, __float_as_int(v)));
}

template <bool kComplexOut, int kV>
__global__ __launch_bounds__(kWaves * 64) void stft1024_hop256_kernel(
    const float* __restrict__ wav, int64_t nsamps, const float* __restrict__ window,
    const float2* __restrict__ twiddle, int64_t nframes, float* __restrict__ out, int pairs_per_clip,
    int64_t total_pairs) {
  constexpr bool kTwoPlanes = (kV & 1) != 0, kTwRegs = (kV & 2) != 0, kWinRegs = (kV & 4) != 0;
  constexpr int kNP = kTwoPlanes ? 2 : 1;
  constexpr int kWinRow = 0, kT1Row = kWinRegs ? 0 : 8, kT2Row = kT1Row + (kTwRegs ? 0 : 8);
  constexpr int kTabRows = kT2Row + (kTwRegs ? 0 : 8);
  __shared__ f2 planes[kWaves][kNP][kPlane];
  __shared__ float2 s_tab[kTabRows ? kTabRows : 1][64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int hi = lane >> 3, lo = lane & 7;
  f2* const plane = &planes[wave][0][0];

  // ---- per-lane constants (they depend on the lane only) ----
  float2 win_r[8], t1_r[8], t2_r[8];
  float tsn[4], tcs[4];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    // window, halved: the 1/2 of the real-FFT split
    const float2 w = *reinterpret_cast<const float2*>(window + 128 * j + 2 * lane);
    const float2 wh = make_float2(0.5f * w.x, 0.5f * w.y);
    // pass-1 twiddle W64^(b*p): lane = (b=hi, c=lo), p = j;  pass-2 twiddle W512^(c*(p+8q)): lane = (p=hi, c=lo), q = j
    const float2 a = twiddle[((hi * j) & 63) * 16];
    const float2 b = twiddle[((lo * (hi + 8 * j)) & 511) * 2];
    if (kWinRegs) win_r[j] = wh; else if (wave == 0) s_tab[kWinRow + j][lane] = wh;
    if (kTwRegs) {
      t1_r[j] = make_float2(a.x, -a.y);
      t2_r[j] = make_float2(b.x, -b.y);
    } else if (wave == 0) {
      s_tab[kT1Row + j][lane] = make_float2(a.x, -a.y);
      s_tab[kT2Row + j][lane] = make_float2(b.x, -b.y);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {   // split twiddle of bin k = lane + 64 r: theta = 2 pi k / 1024
    const float2 c = twiddle[lane + 64 * r];
    tcs[r] = c.x; tsn[r] = c.y;
  }
  if (!(kWinRegs && kTwRegs)) __syncthreads();
  const int partner = ((64 - lane) & 63) * 4;              // ds_bpermute byte index of the lane that holds Z[512 - k]
  f2* const wr = plane + hi * 9 + lo;                      // + 72 p: element (8 p + hi) * 9 + lo
  const unsigned rd1 = advoc::lds_address(plane + 72 * hi + lo);   // + 9 b elements = 72 b bytes
  const unsigned rd2 = advoc::lds_address(plane + 9 * lane);       // + c elements
  constexpr int kIm = kTwoPlanes ? kPlane * 8 : 0;         // byte offset of the imaginary plane

  // Everything that says WHICH pair is wave-uniform and kept in scalar registers (the wave index through readfirstlane);
  // (clip, pair of the clip) advance by a fixed step, so there is no division in the loop and the loads / stores are
  // "scalar base + lane offset" instructions: no 64-bit vector address arithmetic per pair.
  const int swave = __builtin_amdgcn_readfirstlane(wave);
  const int pstride = (int)gridDim.x * kWaves;
  const int step_clip = pstride / pairs_per_clip, step_fp = pstride - step_clip * pairs_per_clip;
  const int pair0 = (int)blockIdx.x * kWaves + swave;
  int clip = pair0 / pairs_per_clip, fp = pair0 - clip * pairs_per_clip;
  const unsigned lane2 = 2u * (unsigned)lane;
  for (int pair = pair0; pair < (int)total_pairs; pair += pstride, clip += step_clip, fp += step_fp) {
    if (fp >= pairs_per_clip) { fp -= pairs_per_clip; ++clip; }
    const int64_t f = 2 * (int64_t)fp;                        // frames f and f + 1
    const bool two = f + 1 < nframes;
    const float* src = wav + (int64_t)clip * nsamps;
    const int64_t s0 = f * 256;
    // samples s0 + 128 j + 2 lane (+ 1), j < 10: frame f is j = 0..7, frame f + 1 is j = 2..9; zero beyond the clip (pad_end)
    float2 raw[10];
    if (s0 + 256 + kNfft <= nsamps) {
      const float* p0 = src + s0;
#pragma unroll
      for (int j = 0; j < 10; ++j) raw[j] = *reinterpret_cast<const float2*>(p0 + (128u * j + lane2));
    } else {
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        const int64_t i0 = s0 + 128 * j + lane2;
        raw[j].x = i0 < nsamps ? src[i0] : 0.f;
        raw[j].y = i0 + 1 < nsamps ? src[i0 + 1] : 0.f;
      }
    }
    f2 re[8], im[8];
    // z[n] = (x[2n] w[2n] + i x[2n+1] w[2n+1]) / 2
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const float2 w = kWinRegs ? win_r[a] : s_tab[kWinRow + a][lane];
      re[a] = f2{raw[a].x * w.x, raw[a + 2].x * w.x};
      im[a] = f2{raw[a].y * w.y, raw[a + 2].y * w.y};
    }

    // pass 1: DFT over a -> p, twiddle, transpose (b,c | p) -> (p,c | b)
    dft8(re, im);
    {
      f2 ti[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        f2 r = re[p], i = im[p];
        if (p) {                      // W64^0 = 1
          const float2 t = kTwRegs ? t1_r[p] : s_tab[kT1Row + p][lane];
          r = re[p] * t.x - im[p] * t.y;
          i = re[p] * t.y + im[p] * t.x;
        }
        wr[72 * p] = r;
        if (kTwoPlanes) wr[72 * p + kPlane] = i; else ti[p] = i;
      }
      ADVOC_LDS_READ8(re, rd1, 0, 72);
      if (kTwoPlanes) {
        ADVOC_LDS_READ8(im, rd1, kIm, 72);
        ADVOC_LDS_WAIT8(re);
        ADVOC_LDS_TIE8(im);
      } else {
        ADVOC_LDS_WAIT8(re);
#pragma unroll
        for (int p = 0; p < 8; ++p) wr[72 * p] = ti[p];
        ADVOC_LDS_READ8(im, rd1, 0, 72);
        ADVOC_LDS_WAIT8(im);
      }
    }

    // pass 2: DFT over b -> q, twiddle, transpose (p,c | q) -> (q,p | c)
    dft8(re, im);
    {
      f2 ti[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float2 t = kTwRegs ? t2_r[q] : s_tab[kT2Row + q][lane];
        const f2 r = re[q] * t.x - im[q] * t.y;
        const f2 i = re[q] * t.y + im[q] * t.x;
        wr[72 * q] = r;
        if (kTwoPlanes) wr[72 * q + kPlane] = i; else ti[q] = i;
      }
      ADVOC_LDS_READ8(re, rd2, 0, 8);
      if (kTwoPlanes) {
        ADVOC_LDS_READ8(im, rd2, kIm, 8);
        ADVOC_LDS_WAIT8(re);
        ADVOC_LDS_TIE8(im);
      } else {
        ADVOC_LDS_WAIT8(re);
#pragma unroll
        for (int q = 0; q < 8; ++q) wr[72 * q] = ti[q];
        ADVOC_LDS_READ8(im, rd2, 0, 8);
        ADVOC_LDS_WAIT8(im);
      }
    }

    // pass 3: DFT over c -> r.  Lane now holds Z[lane + 64 r] / 2.
    dft8(re, im);

    // real-FFT split, two bins per step: with S = Zk + conj(Zm), D = Zk - conj(Zm), m = 512 - k, T = i W1024^k D:
    //   X[k] = S - T,  X[m] = conj(S + T)        (the halves are in the window)
    float* orow0 = out + (((int64_t)clip * nframes + f) * kBins) * (kComplexOut ? 2 : 1);
    float* orow1 = orow0 + kBins * (kComplexOut ? 2 : 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f2 c, d;
      c.x = bperm(partner, re[7 - r].x); c.y = bperm(partner, re[7 - r].y);
      d.x = bperm(partner, im[7 - r].x); d.y = bperm(partner, im[7 - r].y);
      if (lane == 0) {  // k = 64 r pairs with 64 (8 - r) on the same lane (Z[512] = Z[0])
        c = re[(8 - r) & 7];
        d = im[(8 - r) & 7];
      }
      const f2 a = re[r], b = im[r];
      const f2 sr = a + c, si = b - d, dr = a - c, di = b + d;
      const f2 tr = tsn[r] * dr - tcs[r] * di;
      const f2 ti = tsn[r] * di + tcs[r] * dr;
      const f2 xkr = sr - tr, xki = si - ti, xmr = sr + tr, xmi = si + ti;
      const unsigned k = (unsigned)lane + 64u * r, m = 512u - k;
      if (kComplexOut) {
        *reinterpret_cast<float2*>(orow0 + 2 * k) = make_float2(xkr.x, xki.x);
        *reinterpret_cast<float2*>(orow0 + 2 * m) = make_float2(xmr.x, -xmi.x);
        if (two) {
          *reinterpret_cast<float2*>(orow1 + 2 * k) = make_float2(xkr.y, xki.y);
          *reinterpret_cast<float2*>(orow1 + 2 * m) = make_float2(xmr.y, -xmi.y);
        }
      } else {
        const f2 k2 = xkr * xkr + xki * xki, m2 = xmr * xmr + xmi * xmi;
        orow0[k] = __builtin_amdgcn_sqrtf(k2.x);   // v_sqrt_f32, <= 1 ulp
        orow0[m] = __builtin_amdgcn_sqrtf(m2.x);
        if (two) {
          orow1[k] = __builtin_amdgcn_sqrtf(k2.y);
          orow1[m] = __builtin_amdgcn_sqrtf(m2.y);
        }
      }
    }
    if (lane == 0) {  // bin 256 pairs with itself: X[256] = conj(Z[256])
      const f2 xr = 2.0f * re[4], xi = -2.0f * im[4];
      if (kComplexOut) {
        *reinterpret_cast<float2*>(orow0 + 2 * 256) = make_float2(xr.x, xi.x);
        if (two) *reinterpret_cast<float2*>(orow1 + 2 * 256) = make_float2(xr.y, xi.y);
      } else {
        const f2 m2 = xr * xr + xi * xi;
        orow0[256] = __builtin_amdgcn_sqrtf(m2.x);
        if (two) orow1[256] = __builtin_amdgcn_sqrtf(m2.y);
      }
    }
  }
}

// ---- the same kernel with two more cuts in the arithmetic (variant 8) ----
//   * pass 1 runs on (re, im) PAIRS OF ONE FRAME -- an 8-byte load is such a pair, so nothing has to be moved into place
//     (the frame-packed form costs ~38 v_mov per pair to put sample n of frame f next to sample n of frame f + 1) -- and
//     the first transpose writes with ds_write2_b32: the real parts of the two frames from two unrelated registers into
//     one 8-byte slot, i.e. the transposition also converts to the frame-packed form passes 2 and 3 keep (their
//     magnitudes at the end are one packed multiply-add per two bins);
//   * the pass twiddles are applied on the way IN to the next pass, inside its first butterflies: with y = t x4,
//     s = x0 + y and d = 2 x0 - s are six multiply-adds where twiddle + add + subtract are eight.
// The swizzled packed operations (x (-i), complex products) are inline assembly: the compiler builds them from v_mov / v_xor.
#define ADVOC_PK_ADD(D, A, B, MODS) asm("v_pk_add_f32 %0, %1, %2 " MODS : "=v"(D) : "v"(A), "v"(B))
// forward 8-point DFT of complex values held as (re, im) pairs; v[a] <- sum_a v[a] exp(-2 pi i a p / 8)
__device__ __forceinline__ void dft8_cp(f2 (&v)[8]) {
  const float h = 0.70710678118654752440f;
  f2 s[4], d[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    s[n] = v[n] + v[n + 4];
    d[n] = v[n] - v[n + 4];
  }
  {  // d[n] *= W8^n (d[2] x (-i) is folded into its consumers)
    f2 u;
    ADVOC_PK_ADD(u, d[1], d[1], "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]");            // (dr + di, di - dr)
    d[1] = u * h;
    ADVOC_PK_ADD(u, d[3], d[3], "op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[1,1]");   // (di - dr, -dr - di)
    d[3] = u * h;
  }
  {  // even outputs: 4-point DFT of s
    const f2 b0 = s[0] + s[2], b2 = s[0] - s[2], b1 = s[1] + s[3];
    f2 b3;
    ADVOC_PK_ADD(b3, s[1], s[3], "op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[1,0]");  // (s1 - s3) x (-i)
    v[0] = b0 + b1; v[4] = b0 - b1; v[2] = b2 + b3; v[6] = b2 - b3;
  }
  {  // odd outputs: 4-point DFT of (d0, d1, d2 x (-i), d3)
    f2 b0, b2, b3;
    ADVOC_PK_ADD(b0, d[0], d[2], "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]");              // d0 + d2 x (-i)
    ADVOC_PK_ADD(b2, d[0], d[2], "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]");              // d0 - d2 x (-i)
    const f2 b1 = d[1] + d[3];
    ADVOC_PK_ADD(b3, d[1], d[3], "op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[1,0]");
    v[1] = b0 + b1; v[5] = b0 - b1; v[3] = b2 + b3; v[7] = b2 - b3;
  }
}
#undef ADVOC_PK_ADD

// forward 8-point DFT of x[a] = t[a] v[a] (t[0] = 1), frame-packed; TW(a) gives t[a] as float2 {re, im}
template <typename TW>
__device__ __forceinline__ void dft8_tw(f2 (&re)[8], f2 (&im)[8], TW tw) {
  const float h = 0.70710678118654752440f;
  f2 sr[4], si[4], dr[4], di[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    f2 yr = re[n], yi = im[n];
    if (n) {
      const float2 t = tw(n);
      yr = re[n] * t.x - im[n] * t.y;
      yi = re[n] * t.y + im[n] * t.x;
    }
    const float2 u = tw(n + 4);
    sr[n] = yr + re[n + 4] * u.x - im[n + 4] * u.y;        // two multiply-adds each (left to right)
    si[n] = yi + re[n + 4] * u.y + im[n + 4] * u.x;
    dr[n] = 2.0f * yr - sr[n];
    di[n] = 2.0f * yi - si[n];
  }
  {
    f2 r1 = (dr[1] + di[1]) * h, i1 = (di[1] - dr[1]) * h;  // * (1 - i)/sqrt2
    dr[1] = r1; di[1] = i1;
    f2 r2 = di[2], i2 = -dr[2];                             // * (-i)
    dr[2] = r2; di[2] = i2;
    f2 r3 = (di[3] - dr[3]) * h, i3 = -(dr[3] + di[3]) * h; // * (-1 - i)/sqrt2
    dr[3] = r3; di[3] = i3;
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    f2* xr = half ? dr : sr;
    f2* xi = half ? di : si;
    f2 b0r = xr[0] + xr[2], b0i = xi[0] + xi[2];
    f2 b2r = xr[0] - xr[2], b2i = xi[0] - xi[2];
    f2 b1r = xr[1] + xr[3], b1i = xi[1] + xi[3];
    f2 b3r = xi[1] - xi[3], b3i = -(xr[1] - xr[3]);
    re[0 + half] = b0r + b1r; im[0 + half] = b0i + b1i;
    re[4 + half] = b0r - b1r; im[4 + half] = b0i - b1i;
    re[2 + half] = b2r + b3r; im[2 + half] = b2i + b3i;
    re[6 + half] = b2r - b3r; im[6 + half] = b2i - b3i;
  }
}

template <int O0, int O1>
__device__ __forceinline__ void lds_write2_b32(unsigned addr, float a, float b) {   // dword offsets
  asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(addr), "v"(a), "v"(b), "n"(O0), "n"(O1) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_write_b64(unsigned addr, f2 v) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
// rows 8 p + hi, p < 8: element offset 72 p -> byte offset 576 p from the lane's slot (WA: the slot's LDS address)
#define ADVOC_LDS_WRITE8_B64(V, WA)                                                                              \
  {                                                                                                              \
    lds_write_b64<0 * 576>(WA, V[0]); lds_write_b64<1 * 576>(WA, V[1]); lds_write_b64<2 * 576>(WA, V[2]);        \
    lds_write_b64<3 * 576>(WA, V[3]); lds_write_b64<4 * 576>(WA, V[4]); lds_write_b64<5 * 576>(WA, V[5]);        \
    lds_write_b64<6 * 576>(WA, V[6]); lds_write_b64<7 * 576>(WA, V[7]);                                          \
  }
// the same slots from two frames' registers (member M = x | y of Z0[p] and Z1[p]); ds_write2_b32 offsets are dwords < 256
#define ADVOC_LDS_WRITE8_2B32(Z0, Z1, M, WA)                                                                     \
  {                                                                                                              \
    lds_write2_b32<0, 1>(WA, Z0[0].M, Z1[0].M); lds_write2_b32<144, 145>(WA, Z0[1].M, Z1[1].M);                  \
    lds_write2_b32<0, 1>(WA + 1152, Z0[2].M, Z1[2].M); lds_write2_b32<144, 145>(WA + 1152, Z0[3].M, Z1[3].M);    \
    lds_write2_b32<0, 1>(WA + 2304, Z0[4].M, Z1[4].M); lds_write2_b32<144, 145>(WA + 2304, Z0[5].M, Z1[5].M);    \
    lds_write2_b32<0, 1>(WA + 3456, Z0[6].M, Z1[6].M); lds_write2_b32<144, 145>(WA + 3456, Z0[7].M, Z1[7].M);    \
  }

template <bool kComplexOut>
__global__ __launch_bounds__(kWaves * 64) void stft1024_hop256b_kernel(
    const float* __restrict__ wav, int64_t nsamps, const float* __restrict__ window,
    const float2* __restrict__ twiddle, int64_t nframes, float* __restrict__ out, int pairs_per_clip,
    int64_t total_pairs) {
  __shared__ f2 planes[kWaves][kPlane];
  __shared__ float2 s_win[8][64], s_t1[8][64], s_t2[8][64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int hi = lane >> 3, lo = lane & 7;
  f2* const plane = &planes[wave][0];
  float tsn[4], tcs[4];
  if (wave == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float2 w = *reinterpret_cast<const float2*>(window + 128 * j + 2 * lane);
      s_win[j][lane] = make_float2(0.5f * w.x, 0.5f * w.y);              // the 1/2 of the real-FFT split
      // twiddle of input b = j of pass 2, W64^(b p): after the first transpose lane = (p = hi, c = lo)
      const float2 a = twiddle[((hi * j) & 63) * 16];
      s_t1[j][lane] = make_float2(a.x, -a.y);
      // twiddle of input c = j of pass 3, W512^(c (p + 8 q)): after the second transpose lane = 8 q + p
      const float2 b = twiddle[((lane * j) & 511) * 2];
      s_t2[j][lane] = make_float2(b.x, -b.y);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float2 c = twiddle[lane + 64 * r];
    tcs[r] = c.x; tsn[r] = c.y;
  }
  __syncthreads();
  const int partner = ((64 - lane) & 63) * 4;
  const unsigned wa = advoc::lds_address(plane + hi * 9 + lo);
  const unsigned rd1 = advoc::lds_address(plane + 72 * hi + lo);
  const unsigned rd2 = advoc::lds_address(plane + 9 * lane);

  const int swave = __builtin_amdgcn_readfirstlane(wave);
  const int pstride = (int)gridDim.x * kWaves;
  const int step_clip = pstride / pairs_per_clip, step_fp = pstride - step_clip * pairs_per_clip;
  const int pair0 = (int)blockIdx.x * kWaves + swave;
  int clip = pair0 / pairs_per_clip, fp = pair0 - clip * pairs_per_clip;
  const unsigned lane2 = 2u * (unsigned)lane;
  for (int pair = pair0; pair < (int)total_pairs; pair += pstride, clip += step_clip, fp += step_fp) {
    if (fp >= pairs_per_clip) { fp -= pairs_per_clip; ++clip; }
    const int64_t f = 2 * (int64_t)fp;
    const bool two = f + 1 < nframes;
    const float* src = wav + (int64_t)clip * nsamps;
    const int64_t s0 = f * 256;
    f2 raw[10];
    if (s0 + 256 + kNfft <= nsamps) {
      const float* p0 = src + s0;
#pragma unroll
      for (int j = 0; j < 10; ++j) raw[j] = *reinterpret_cast<const f2*>(p0 + (128u * j + lane2));
    } else {
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        const int64_t i0 = s0 + 128 * j + lane2;
        raw[j].x = i0 < nsamps ? src[i0] : 0.f;
        raw[j].y = i0 + 1 < nsamps ? src[i0 + 1] : 0.f;
      }
    }
    f2 re[8], im[8];
    {
      // pass 1 on (re, im) pairs: z[n] = (x[2n] w[2n], x[2n+1] w[2n+1]) / 2, n = 64 a + lane; DFT over a -> p
      f2 z0[8], z1[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const float2 w2 = s_win[a][lane];
        const f2 w = {w2.x, w2.y};
        z0[a] = raw[a] * w;
        z1[a] = raw[a + 2] * w;
      }
      dft8_cp(z0);
      dft8_cp(z1);
      // transpose (b,c | p) -> (p,c | b), real parts then imaginary parts through the one plane; the slots come back frame-packed
      ADVOC_LDS_WRITE8_2B32(z0, z1, x, wa);
      ADVOC_LDS_READ8(re, rd1, 0, 72);
      ADVOC_LDS_WAIT8(re);
      ADVOC_LDS_WRITE8_2B32(z0, z1, y, wa);
      ADVOC_LDS_READ8(im, rd1, 0, 72);
      ADVOC_LDS_WAIT8(im);
    }
    // pass 2: twiddle W64^(b p) and DFT over b -> q, transpose (p,c | q) -> (q,p | c)
    dft8_tw(re, im, [&](int b) { return s_t1[b][lane]; });
    {
      f2 ti[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) ti[q] = im[q];
      ADVOC_LDS_WRITE8_B64(re, wa);
      ADVOC_LDS_READ8(re, rd2, 0, 8);
      ADVOC_LDS_WAIT8(re);
      ADVOC_LDS_WRITE8_B64(ti, wa);
      ADVOC_LDS_READ8(im, rd2, 0, 8);
      ADVOC_LDS_WAIT8(im);
    }
    // pass 3: twiddle W512^(c (p + 8 q)) and DFT over c -> r.  Lane now holds Z[lane + 64 r] / 2.
    dft8_tw(re, im, [&](int c) { return s_t2[c][lane]; });

    float* orow0 = out + (((int64_t)clip * nframes + f) * kBins) * (kComplexOut ? 2 : 1);
    float* orow1 = orow0 + kBins * (kComplexOut ? 2 : 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f2 c, d;
      c.x = bperm(partner, re[7 - r].x); c.y = bperm(partner, re[7 - r].y);
      d.x = bperm(partner, im[7 - r].x); d.y = bperm(partner, im[7 - r].y);
      if (lane == 0) {
        c = re[(8 - r) & 7];
        d = im[(8 - r) & 7];
      }
      const f2 a = re[r], b = im[r];
      const f2 sr = a + c, si = b - d, dr = a - c, di = b + d;
      const f2 tr = tsn[r] * dr - tcs[r] * di;
      const f2 ti = tsn[r] * di + tcs[r] * dr;
      const f2 xkr = sr - tr, xki = si - ti, xmr = sr + tr, xmi = si + ti;
      const unsigned k = (unsigned)lane + 64u * r, m = 512u - k;
      if (kComplexOut) {
        *reinterpret_cast<float2*>(orow0 + 2 * k) = make_float2(xkr.x, xki.x);
        *reinterpret_cast<float2*>(orow0 + 2 * m) = make_float2(xmr.x, -xmi.x);
        if (two) {
          *reinterpret_cast<float2*>(orow1 + 2 * k) = make_float2(xkr.y, xki.y);
          *reinterpret_cast<float2*>(orow1 + 2 * m) = make_float2(xmr.y, -xmi.y);
        }
      } else {
        const f2 k2 = xkr * xkr + xki * xki, m2 = xmr * xmr + xmi * xmi;
        orow0[k] = __builtin_amdgcn_sqrtf(k2.x);
        orow0[m] = __builtin_amdgcn_sqrtf(m2.x);
        if (two) {
          orow1[k] = __builtin_amdgcn_sqrtf(k2.y);
          orow1[m] = __builtin_amdgcn_sqrtf(m2.y);
        }
      }
    }
    if (lane == 0) {
      const f2 xr = 2.0f * re[4], xi = -2.0f * im[4];
      if (kComplexOut) {
        *reinterpret_cast<float2*>(orow0 + 2 * 256) = make_float2(xr.x, xi.x);
        if (two) *reinterpret_cast<float2*>(orow1 + 2 * 256) = make_float2(xr.y, xi.y);
      } else {
        const f2 m2 = xr * xr + xi * xi;
        orow0[256] = __builtin_amdgcn_sqrtf(m2.x);
        if (two) orow1[256] = __builtin_amdgcn_sqrtf(m2.y);
      }
    }
  }
}
#undef ADVOC_LDS_WRITE8_B64
#undef ADVOC_LDS_WRITE8_2B32
#undef ADVOC_LDS_READ8
#undef ADVOC_LDS_WAIT8
#undef ADVOC_LDS_TIE8

// ---------------------------------------------------------------------------------------------
// Inverse: one wavefront = one frame.  irfft(1024) runs as the SAME 512-point complex transform
// (IDFT(Z) = conj(DFT(conj(Z))) / 512) after undoing the real-FFT split:
//   Ze[k] = (X[k] + conj(X[512-k])) / 2,  Zo[k] = (X[k] - conj(X[512-k])) / 2 * W1024^-k,
//   Z[k] = Ze[k] + i Zo[k];   x[2n] + i x[2n+1] = IDFT512(Z)[n].
// The frame is multiplied by the synthesis window and written to `frames` ([batch, T, 1024]);
// ola_kernel then sums the <= nfft/hop frames that cover each output sample (a gather: no atomics).
// Replaces lws.istft reached from advoc/spectral.py:300-309,320-321.
// ---------------------------------------------------------------------------------------------
// kProject: the Griffin-Lim projection fused into the loads -- every bin is replaced by
// |mag| * X / |X| (phase 0 where X == 0) on its way in, so the projected spectrum never exists in HBM.
template <bool kProject>
__global__ __launch_bounds__(kWaves * 64) void istft1024_frames_kernel(
    const float2* __restrict__ spec, const float* __restrict__ mag, const float* __restrict__ window,
    const float2* __restrict__ twiddle, int64_t total_frames, float* __restrict__ frames) {
  __shared__ float planes[kWaves][2 * kPlane];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float* xr_plane = &planes[wave][0];
  float* xi_plane = xr_plane + kPlane;
  const int hi = lane >> 3, lo = lane & 7;
  float w0[8], w1[8];
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    const float2 w = *reinterpret_cast<const float2*>(window + 128 * a + 2 * lane);
    w0[a] = w.x;
    w1[a] = w.y;
  }
  float t1r[8], t1i[8], t2r[8], t2i[8], tsn[8], tcs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float2 a = twiddle[((hi * j) & 63) * 16];
    t1r[j] = a.x; t1i[j] = -a.y;
    const float2 b = twiddle[((lo * (hi + 8 * j)) & 511) * 2];
    t2r[j] = b.x; t2i[j] = -b.y;
    const float2 c = twiddle[lane + 64 * j];
    tcs[j] = c.x; tsn[j] = c.y;
  }
  for (int64_t f = (int64_t)blockIdx.x * kWaves + wave; f < total_frames; f += (int64_t)gridDim.x * kWaves) {
    const float2* X = spec + f * kBins;
    float re[8], im[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const int k = 64 * a + lane;
      float2 xk = X[k];
      float2 xm = X[512 - k];
      if (kProject) {
        const float mk = fabsf(mag[f * kBins + k]), mm = fabsf(mag[f * kBins + 512 - k]);
        const float ak = sqrtf(xk.x * xk.x + xk.y * xk.y), am = sqrtf(xm.x * xm.x + xm.y * xm.y);
        xk = ak > 0.f ? make_float2(mk * (xk.x / ak), mk * (xk.y / ak)) : make_float2(mk, 0.f);
        xm = am > 0.f ? make_float2(mm * (xm.x / am), mm * (xm.y / am)) : make_float2(mm, 0.f);
      }
      if (k == 0) { xk.y = 0.f; xm.y = 0.f; }      // irfft ignores Im X[0] and Im X[N/2]
      const float er = 0.5f * (xk.x + xm.x), ei = 0.5f * (xk.y - xm.y);      // Ze
      const float dr = 0.5f * (xk.x - xm.x), di = 0.5f * (xk.y + xm.y);      // (X[k] - conj X[m]) / 2
      const float zor = dr * tcs[a] - di * tsn[a], zoi = dr * tsn[a] + di * tcs[a];   // * e^{+i theta_k}
      re[a] = er - zoi;                 // Z = Ze + i Zo
      im[a] = -(ei + zor);              // conjugated for the forward machinery
    }
    dft8(re, im);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const float r = re[p] * t1r[p] - im[p] * t1i[p];
      const float i = re[p] * t1i[p] + im[p] * t1r[p];
      const int addr = (8 * p + hi) * 9 + lo;
      xr_plane[addr] = r;
      xi_plane[addr] = i;
    }
    wave_lds_sync();
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int addr = (8 * hi + b) * 9 + lo;
      re[b] = xr_plane[addr];
      im[b] = xi_plane[addr];
    }
    wave_lds_sync();
    dft8(re, im);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float r = re[q] * t2r[q] - im[q] * t2i[q];
      const float i = re[q] * t2i[q] + im[q] * t2r[q];
      const int addr = (8 * q + hi) * 9 + lo;
      xr_plane[addr] = r;
      xi_plane[addr] = i;
    }
    wave_lds_sync();
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int addr = lane * 9 + c;
      re[c] = xr_plane[addr];
      im[c] = xi_plane[addr];
    }
    wave_lds_sync();
    dft8(re, im);
    // lane holds conj(z[n]) * 512 for n = lane + 64 r:  x[2n] = re / 512, x[2n+1] = -im / 512
    float* orow = frames + f * kNfft;
    const float sc = 1.0f / 512.0f;
#pragma unroll
    for (int r = 0; r < 8; ++r)
      *reinterpret_cast<float2*>(orow + 128 * r + 2 * lane) = make_float2(re[r] * sc * w0[r], -im[r] * sc * w1[r]);
  }
}

// out[clip][s] = sum over frames f with f*hop <= s < f*hop + nfft of frames[clip][f][s - f*hop]
__global__ __launch_bounds__(256) void ola_kernel(const float* __restrict__ frames, int64_t batch, int64_t nframes,
                                                  int nhop, int64_t out_len, float* __restrict__ out) {
  const int64_t quads = out_len / 4;     // out_len = (T-1)*hop + 1024, hop % 4 == 0
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= batch * quads) return;
  const int64_t clip = i / quads;
  const int64_t s = (i - clip * quads) * 4;
  int64_t f_hi = s / nhop;
  if (f_hi > nframes - 1) f_hi = nframes - 1;
  int64_t f_lo = (s - kNfft + nhop) / nhop;      // smallest f with f*hop + nfft > s  (s+3 shares it: hop % 4 == 0)
  if (s - kNfft + nhop < 0) f_lo = 0;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* base = frames + clip * nframes * kNfft;
  for (int64_t f = f_lo; f <= f_hi; ++f) {
    const float4 v = *reinterpret_cast<const float4*>(base + f * kNfft + (s - f * nhop));
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  *reinterpret_cast<float4*>(out + clip * out_len + s) = acc;
}

// spec <- mag * spec / |spec|  (|spec| == 0 -> phase 0, as np.angle(0) == 0)   advoc/spectral.py:306-307
__global__ __launch_bounds__(256) void phase_project_kernel(float2* __restrict__ spec, const float* __restrict__ mag,
                                                            int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float2 x = spec[i];
  const float m = fabsf(mag[i]);
  const float a = sqrtf(x.x * x.x + x.y * x.y);
  spec[i] = a > 0.f ? make_float2(m * (x.x / a), m * (x.y / a)) : make_float2(m, 0.f);
}

// out[i] = |spec[i]|  (tf.abs of a complex64 tensor, advoc/loader.py:128 / spectral.py:203)
__global__ __launch_bounds__(256) void cabs_kernel(const float2* __restrict__ spec, float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float2 x = spec[i];
  out[i] = __builtin_amdgcn_sqrtf(x.x * x.x + x.y * x.y);
}

// spec = |mag| * exp(2 pi i u)   advoc/spectral.py:301-304
__global__ __launch_bounds__(256) void polar_kernel(const float* __restrict__ mag, const float* __restrict__ u,
                                                    float2* __restrict__ spec, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float sn, cs;
  sincospif(2.0f * u[i], &sn, &cs);
  const float m = fabsf(mag[i]);
  spec[i] = make_float2(m * cs, m * sn);
}

int launch_stft(const float* wav, int64_t batch, int64_t nsamps, const float* window,
                const float* twiddle, int32_t nfft, int32_t nhop, int64_t nframes, float* out,
                bool complex_out, hipStream_t stream) {
  if (batch < 0 || nsamps < 0 || nframes < 0 || nhop <= 0) return ADVOC_ERR_BAD_SHAPE;
  if (nfft != kNfft || (nhop & 1) || nhop > 4096) return ADVOC_ERR_UNSUPPORTED;
  if (batch == 0 || nframes == 0) return ADVOC_OK;  // empty output: nothing to touch
  if (!wav || !window || !twiddle || !out) return ADVOC_ERR_NULL;
  if (reinterpret_cast<uintptr_t>(wav) & 7) return ADVOC_ERR_UNSUPPORTED;    // 8-byte sample loads
  const int64_t pairs_per_clip = (nframes + 1) / 2;
  const int64_t total_pairs = batch * pairs_per_clip;
  if (pairs_per_clip > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
  // a wave keeps ~64 constant registers (window + twiddles): grid-stride so that set-up is
  // amortised; 8 workgroups per CU is more than the register file holds at once
  int64_t blocks = advoc::ceil_div(total_pairs, kWaves);
  if (blocks > 2048) blocks = 2048;
  // hop 256 with 8-byte aligned clips: the kernel with fewer instructions per frame pair
  static const int variant = getenv("ADVOC_STFT_V") ? atoi(getenv("ADVOC_STFT_V")) : 0;
  constexpr int block_cap = 1024;   // four workgroups per CU, resident
  if (nhop == 256 && !(nsamps & 1) && total_pairs < (1LL << 30) && variant >= 0) {
    if (blocks > block_cap) blocks = block_cap;
#define ADVOC_STFT_LAUNCH(C, V)                                                                                         \
  hipLaunchKernelGGL((stft1024_hop256_kernel<C, V>), dim3((unsigned)blocks), dim3(kWaves * 64), 0, stream, wav, nsamps, \
                     window, reinterpret_cast<const float2*>(twiddle), nframes, out, (int)pairs_per_clip, total_pairs)
#define ADVOC_STFT_CASE(V) case V: if (complex_out) ADVOC_STFT_LAUNCH(true, V); else ADVOC_STFT_LAUNCH(false, V); break
    ADVOC_CLEAR_LAUNCH_ERROR();
    switch (variant) {
      ADVOC_STFT_CASE(0); ADVOC_STFT_CASE(1);
      case 8:
        if (complex_out)
          hipLaunchKernelGGL(stft1024_hop256b_kernel<true>, dim3((unsigned)blocks), dim3(kWaves * 64), 0, stream, wav, nsamps, window,
                             reinterpret_cast<const float2*>(twiddle), nframes, out, (int)pairs_per_clip, total_pairs);
        else
          hipLaunchKernelGGL(stft1024_hop256b_kernel<false>, dim3((unsigned)blocks), dim3(kWaves * 64), 0, stream, wav, nsamps, window,
                             reinterpret_cast<const float2*>(twiddle), nframes, out, (int)pairs_per_clip, total_pairs);
        break;
      default: return ADVOC_ERR_UNSUPPORTED;
    }
#undef ADVOC_STFT_CASE
#undef ADVOC_STFT_LAUNCH
    ADVOC_RETURN_IF_LAUNCH_FAILED();
    return ADVOC_OK;
  }
  if (complex_out) {
    ADVOC_CLEAR_LAUNCH_ERROR();
    hipLaunchKernelGGL(stft1024_kernel<true>, dim3((unsigned)blocks), dim3(kWaves * 64), 0, stream, wav, nsamps,
                       window, reinterpret_cast<const float2*>(twiddle), nhop, nframes, out, (int)pairs_per_clip,
                       total_pairs);
  } else {
    ADVOC_CLEAR_LAUNCH_ERROR();
    hipLaunchKernelGGL(stft1024_kernel<false>, dim3((unsigned)blocks), dim3(kWaves * 64), 0, stream, wav, nsamps,
                       window, reinterpret_cast<const float2*>(twiddle), nhop, nframes, out, (int)pairs_per_clip,
                       total_pairs);
  }
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

}  // namespace

extern "C" int advoc_stft_mag_f32(const float* wav, int64_t batch, int64_t nsamps,
                                  const float* window, const float* twiddle, int32_t nfft,
                                  int32_t nhop, int64_t nframes, float* mag, advoc_stream_t stream) {
  return launch_stft(wav, batch, nsamps, window, twiddle, nfft, nhop, nframes, mag, false,
                     advoc::as_stream(stream));
}

extern "C" int advoc_stft_c64(const float* wav, int64_t batch, int64_t nsamps, const float* window,
                              const float* twiddle, int32_t nfft, int32_t nhop, int64_t nframes,
                              float* out, advoc_stream_t stream) {
  return launch_stft(wav, batch, nsamps, window, twiddle, nfft, nhop, nframes, out, true,
                     advoc::as_stream(stream));
}

namespace {
int launch_istft(const float* spec, const float* mag, int64_t batch, int64_t nframes, const float* window,
                 const float* twiddle, int32_t nfft, int32_t nhop, float* frames_work, float* wav,
                 hipStream_t stream) {
  if (batch < 0 || nframes < 0 || nhop <= 0) return ADVOC_ERR_BAD_SHAPE;
  if (nfft != kNfft || (nhop & 3) || nhop > kNfft) return ADVOC_ERR_UNSUPPORTED;
  if (batch == 0 || nframes == 0) return ADVOC_OK;
  if (!spec || !window || !twiddle || !frames_work || !wav) return ADVOC_ERR_NULL;
  const int64_t total = batch * nframes;
  int64_t blocks = advoc::ceil_div(total, kWaves);
  if (blocks > 4096) blocks = 4096;
  ADVOC_CLEAR_LAUNCH_ERROR();
  if (mag)
    hipLaunchKernelGGL(istft1024_frames_kernel<true>, dim3((unsigned)blocks), dim3(kWaves * 64), 0, stream,
                       reinterpret_cast<const float2*>(spec), mag, window, reinterpret_cast<const float2*>(twiddle),
                       total, frames_work);
  else
    hipLaunchKernelGGL(istft1024_frames_kernel<false>, dim3((unsigned)blocks), dim3(kWaves * 64), 0, stream,
                       reinterpret_cast<const float2*>(spec), mag, window, reinterpret_cast<const float2*>(twiddle),
                       total, frames_work);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  const int64_t out_len = (nframes - 1) * nhop + kNfft;
  const int64_t threads = batch * (out_len / 4);
  const int64_t ob = advoc::ceil_div(threads, 256);
  if (ob > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(ola_kernel, dim3((unsigned)ob), dim3(256), 0, stream, frames_work, batch, nframes, nhop,
                     out_len, wav);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}
}  // namespace

extern "C" int advoc_istft_f32(const float* spec, int64_t batch, int64_t nframes, const float* window,
                               const float* twiddle, int32_t nfft, int32_t nhop, float* frames_work,
                               float* wav, advoc_stream_t stream) {
  return launch_istft(spec, nullptr, batch, nframes, window, twiddle, nfft, nhop, frames_work, wav,
                      advoc::as_stream(stream));
}

extern "C" int advoc_istft_project_f32(const float* spec, const float* mag, int64_t batch, int64_t nframes,
                                       const float* window, const float* twiddle, int32_t nfft, int32_t nhop,
                                       float* frames_work, float* wav, advoc_stream_t stream) {
  if (!mag && batch > 0 && nframes > 0) return ADVOC_ERR_NULL;
  return launch_istft(spec, mag, batch, nframes, window, twiddle, nfft, nhop, frames_work, wav,
                      advoc::as_stream(stream));
}

extern "C" int advoc_phase_project_c64(float* spec, const float* mag, int64_t n, advoc_stream_t stream) {
  if (n < 0) return ADVOC_ERR_BAD_SHAPE;
  if (n == 0) return ADVOC_OK;
  if (!spec || !mag) return ADVOC_ERR_NULL;
  const int64_t blocks = advoc::ceil_div(n, 256);
  if (blocks > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(phase_project_kernel, dim3((unsigned)blocks), dim3(256), 0, advoc::as_stream(stream),
                     reinterpret_cast<float2*>(spec), mag, n);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

extern "C" int advoc_cabs_f32(const float* spec, float* out, int64_t n, advoc_stream_t stream) {
  if (n < 0) return ADVOC_ERR_BAD_SHAPE;
  if (n == 0) return ADVOC_OK;
  if (!spec || !out) return ADVOC_ERR_NULL;
  const int64_t blocks = advoc::ceil_div(n, 256);
  if (blocks > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(cabs_kernel, dim3((unsigned)blocks), dim3(256), 0, advoc::as_stream(stream),
                     reinterpret_cast<const float2*>(spec), out, n);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

extern "C" int advoc_polar_c64(const float* mag, const float* unit_phase, float* spec, int64_t n,
                               advoc_stream_t stream) {
  if (n < 0) return ADVOC_ERR_BAD_SHAPE;
  if (n == 0) return ADVOC_OK;
  if (!spec || !mag || !unit_phase) return ADVOC_ERR_NULL;
  const int64_t blocks = advoc::ceil_div(n, 256);
  if (blocks > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(polar_kernel, dim3((unsigned)blocks), dim3(256), 0, advoc::as_stream(stream), mag,
                     unit_phase, reinterpret_cast<float2*>(spec), n);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

// Fills the twiddle table the STFT kernels read: tw[2e] = cos(2 pi e / nfft), tw[2e+1] =
// sin(2 pi e / nfft), e in [0, nfft), evaluated in double precision on the HOST.
extern "C" int advoc_stft_twiddle_host(float* tw_host, int32_t nfft) {
  if (!tw_host) return ADVOC_ERR_NULL;
  if (nfft != kNfft) return ADVOC_ERR_UNSUPPORTED;
  const double two_pi = 6.283185307179586476925286766559;
  for (int e = 0; e < nfft; ++e) {
    tw_host[2 * e] = (float)cos(two_pi * e / nfft);
    tw_host[2 * e + 1] = (float)sin(two_pi * e / nfft);
  }
  return ADVOC_OK;
}
