// 1024-point real FFT of TWO frames per wavefront (packed fp32) as inline pieces for the fused extractor (extract.hip).
// Same arithmetic as stft1024_kernel (stft.hip, which keeps its own inlined copy: routed through these functions it
// went from 126 registers to 128 + 80 bytes of scratch, and it runs at four waves per SIMD).  See stft.hip for the mapping
// to the hardware.
#pragma once
#include "common.h"
#include "lds_dma.h"

namespace advoc {
namespace fft1024 {

constexpr int kNfft = 1024;
constexpr int kBins = kNfft / 2 + 1;
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

// The window and the pass-1 / pass-2 twiddles of the 512-point complex FFT, [j][lane] in LDS (they depend on the lane
// only); filled by ONE wave of the workgroup, then a workgroup barrier
struct Tables {
  float2 win[8][64], t1[8][64], t2[8][64];
};
__device__ __forceinline__ void fill_tables(Tables& t, const float* __restrict__ window, const float2* __restrict__ twiddle,
                                            int lane) {
  const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    t.win[j][lane] = *reinterpret_cast<const float2*>(window + 128 * j + 2 * lane);
    // pass-1 twiddle W64^(b*p): lane = (b=hi, c=lo), p = j
    const float2 a = twiddle[((hi * j) & 63) * 16];
    t.t1[j][lane] = make_float2(a.x, -a.y);
    // pass-2 twiddle W512^(c*(p+8q)): lane = (p=hi, c=lo), q = j
    const float2 b = twiddle[((lo * (hi + 8 * j)) & 511) * 2];
    t.t2[j][lane] = make_float2(b.x, -b.y);
  }
}

// raw0 / raw1: samples x[2n], x[2n+1], n = 64 a + lane, of the two frames -> Z[lane + 64 r] of the 512-point complex FFT
// of z[n] = x[2n] w[2n] + i x[2n+1] w[2n+1] in (re[r], im[r]).  `plane`: kPlane f2 of LDS private to the wave.
__device__ __forceinline__ void forward_pair(const float2 (&raw0)[8], const float2 (&raw1)[8], const Tables& tb,
                                             f2* __restrict__ plane, int lane, f2 (&re)[8], f2 (&im)[8]) {
  const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    const float2 w = tb.win[a][lane];
    re[a] = f2{raw0[a].x, raw1[a].x} * w.x;
    im[a] = f2{raw0[a].y, raw1[a].y} * w.y;
  }
  // pass 1: DFT over a -> p, twiddle, transpose (b,c | p) -> (p,c | b)
  dft8(re, im);
  {
    f2 ti[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const float2 t = tb.t1[p][lane];
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
      const float2 t = tb.t2[q][lane];
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
}

// real-FFT split of bin k = lane + 64 r: X[k] = ((Zk + conj(Zm)) - i W1024^k (Zk - conj(Zm))) / 2, m = 512 - k
// (tcs / tsn: cos / sin(2 pi k / 1024) of this lane's bins)
__device__ __forceinline__ void split_bin(const f2 (&re)[8], const f2 (&im)[8], int r, int lane, float tcs_r, float tsn_r,
                                          f2& xr, f2& xi) {
  const int partner = (64 - lane) & 63;
  f2 c, d;
  c.x = __shfl(re[7 - r].x, partner, 64); c.y = __shfl(re[7 - r].y, partner, 64);
  d.x = __shfl(im[7 - r].x, partner, 64); d.y = __shfl(im[7 - r].y, partner, 64);
  if (lane == 0) {  // k = 64 r pairs with 512 - 64 r = 64 (8 - r) on the same lane
    c = re[(8 - r) & 7];
    d = im[(8 - r) & 7];
  }
  const f2 a = re[r], b = im[r];
  const f2 sr = a + c, si = b - d, dr = a - c, di = b + d;
  xr = 0.5f * (sr - (tsn_r * dr - tcs_r * di));
  xi = 0.5f * (si - (tsn_r * di + tcs_r * dr));
}


// ---- pieces of the hop-256 form (stft1024_hop256_kernel in stft.hip explains them) ----
// ds_read_b64 the load/store optimiser does not fuse into ds_read2_b64 (8 LDS cycles per 1 KB against 2 per 512 B)
template <int OFF>
__device__ __forceinline__ f2 lds_read_b64(unsigned addr) {
  f2 v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
#define ADVOC_FFT_LDS_READ8(DST, ADDR, BASE, STEP)                                                                     \
  {                                                                                                                    \
    DST[0] = advoc::fft1024::lds_read_b64<(BASE) + 0 * (STEP)>(ADDR); DST[1] = advoc::fft1024::lds_read_b64<(BASE) + 1 * (STEP)>(ADDR); \
    DST[2] = advoc::fft1024::lds_read_b64<(BASE) + 2 * (STEP)>(ADDR); DST[3] = advoc::fft1024::lds_read_b64<(BASE) + 3 * (STEP)>(ADDR); \
    DST[4] = advoc::fft1024::lds_read_b64<(BASE) + 4 * (STEP)>(ADDR); DST[5] = advoc::fft1024::lds_read_b64<(BASE) + 5 * (STEP)>(ADDR); \
    DST[6] = advoc::fft1024::lds_read_b64<(BASE) + 6 * (STEP)>(ADDR); DST[7] = advoc::fft1024::lds_read_b64<(BASE) + 7 * (STEP)>(ADDR); \
  }
// every LDS read of this wave has returned; the eight values are "written" by the statement, so nothing that uses them can
// be scheduled above it
#define ADVOC_FFT_LDS_WAIT8(V)                                                                                      \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                               \
               : "+v"(V[0]), "+v"(V[1]), "+v"(V[2]), "+v"(V[3]), "+v"(V[4]), "+v"(V[5]), "+v"(V[6]), "+v"(V[7])     \
               :: "memory")
__device__ __forceinline__ float bperm(int byte_index, float v) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_index, __float_as_int(v)));
}

}  // namespace fft1024
}  // namespace advoc
