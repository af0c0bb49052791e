// The feature extractor of a training batch in ONE launch: waveform -> |STFT| -> mel -> pseudo-inverse magnitudes
// (advoc/loader.py:116-128 `tf.abs(stft_tf(...))`, then models/advoc/spectral_util.py:29-43 as
// models/advoc/train_evaluate.py:55-56 chains them; SURVEY.md §8b-7 `stft_mel_f32`).  The three outputs are written
// once (1 397 760 B per 256-frame clip with the 265 216 B of samples read once); the magnitudes are never re-read.
//
// The two kernels this replaces are bound by different things: stft1024_kernel by VALU / LDS issue (a 1024-point FFT per
// 3 KB), mel_pinv_kernel by HBM (4.4 KB per frame).  Run back to back neither overlaps the other.  Here a workgroup
// (8 waves, 65 KB of LDS -> TWO per CU) owns tiles of 16 consecutive frames of one clip and alternates
//   FFT    each wave transforms its frame pair (fft1024.h), stores the 2 x 513 magnitudes and leaves them in its LDS plane;
//   mel    the filterbank's runs of non-zero weights on the vector ALUs, straight from the eight planes;
//   pinv   [16 x 80] x [80 x 513] on the f16 matrix cores (32x32x16 tiles whose rows 16..31 are zero: the matrix pipes are
//          idle anyway) with melpinv.hip's arithmetic -- mel rows as fp16 pairs under their own power-of-two scale; the
//          pseudo-inverse arrives as a PRE-SPLIT fp16-pair table in MFMA operand order (built once on the host,
//          170 KB, L2 resident: 16-byte loads, no conversion, no registers held across tiles);
// so that on a CU one workgroup's store-heavy phases run next to the other's FFT.  Arithmetic and outputs are those of
// stft1024_kernel followed by mel_pinv_kernel (tests/test_hip_spectral.py compares the two paths).
//
// Measured (r3, tools/micro/extract_time.py, profiles/r03_extract_sq.md): 76 us for the 128-clip training feed (two
// launches: 86 us) but 335 us for 512 clips (two launches: 277 us): the cycle count of the fused launch is the SUM of the
// two kernels' (+5 %), not their maximum -- both halves are issue-bound (mel_pinv_kernel executes as many vector
// instructions as the FFT), the workgroup's two barriers per tile park its waves 60 % of the time, and the pre-split
// pseudo-inverse is 170 KB of L2 reads per 16-frame tile.  The host side therefore takes this kernel for launches of up to
// 65 536 frames (where the two-launch path leaves the chip half empty between launches) and the two kernels above that.
// What did NOT help: one workgroup of 256 registers per CU (420 us), a half-tile start stagger of the second workgroup
// (no change), routing the FFT through fft1024.h's functions (+22 registers: 400 B of scratch at 128).
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include "common.h"
#include "fft1024.h"

namespace {

using namespace advoc::fft1024;

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kXWaves = 8, kXThreads = 64 * kXWaves;
constexpr int kTile = 2 * kXWaves;          // frames per tile
constexpr int kXPlane = 578;                // f2 per wave plane: 513 magnitudes (>= the FFT's 576); consecutive planes are
                                            // 16 bytes apart modulo the 128-byte bank row, so that the same bin of eight
                                            // planes is eight different banks
constexpr int kMels = 80, kMelPitch = 81, kNBlocks = 17 /* 32-bin blocks of the pre-split table; the last holds bin 512 only */, kSteps = kMels / 16, kMaxW = 2048;

struct XSmem {
  Tables tables;                            // 12 288 B
  float2 split[8][64];                      //  4 096 B  cos / sin(2 pi k / 1024), k = lane + 64 j
  f2 planes[kXWaves][kXPlane];              // 36 992 B
  float w[kMaxW];                           //  8 192 B  packed filterbank runs
  int band[3 * kMels];                      //    960 B  first bin, run length / 4, offset into w
  float mel[kTile * kMelPitch];             //  5 184 B
  float p512[kMels];                        //    320 B  pseudo-inverse row of bin 512 (fp32), see the pinv phase
};

__device__ __forceinline__ float pair_scale(float amax) {
  const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu);
  if (e == 0 || e == 255) return 1.f;
  int sh = 13 - (e - 127);
  sh = sh > 120 ? 120 : (sh < -120 ? -120 : sh);
  return __uint_as_float((unsigned)(sh + 127) << 23);
}
__device__ __forceinline__ void pair_split8(const float* v, float up, f16x8& h0, f16x8& h1) {
  __half2 p0[4], p1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = v[2 * i] * up, b = v[2 * i + 1] * up;
    const __half a0 = __float2half_rn(a), b0 = __float2half_rn(b);
    p0[i] = __halves2half2(a0, b0);
    p1[i] = __halves2half2(__float2half_rn(a - __half2float(a0)), __float2half_rn(b - __half2float(b0)));
  }
  h0 = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(p0));
  h1 = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(p1));
}

template <int MINW>     // waves per SIMD the register allocation aims at: 4 = two workgroups per CU, 2 = one
__global__ __launch_bounds__(kXThreads, MINW) void stft_mel_pinv_kernel(
    const float* __restrict__ wav, int64_t nsamps, const float* __restrict__ window, const float2* __restrict__ twiddle,
    int nhop, int T, const float* __restrict__ mel_wp, const int2* __restrict__ band, int packed,
    const uint4* __restrict__ pinv_pairs, const float* __restrict__ pinv_unscale, float* __restrict__ mag_out,
    float* __restrict__ mel_out, float* __restrict__ inv_out, int tiles_per_clip, int total_tiles, int skew_sleeps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char xsm_raw[];
  XSmem& sm = *reinterpret_cast<XSmem*>(xsm_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f2* plane = &sm.planes[wave][0];

  // ---- once per workgroup ----
  if (wave == 0) {
    fill_tables(sm.tables, window, twiddle, lane);
#pragma unroll
    for (int j = 0; j < 8; ++j) {        // the 1/2 of the real-FFT split, in the window
      const float2 w = sm.tables.win[j][lane];
      sm.tables.win[j][lane] = make_float2(0.5f * w.x, 0.5f * w.y);
    }
  }
  for (int i = tid; i < packed; i += kXThreads) sm.w[i] = mel_wp[i];
  if (tid < kMels) {
    const int2 b = band[tid];
    sm.band[tid] = b.x;
    sm.band[kMels + tid] = (b.y - b.x + 3) >> 2;       // runs are padded to multiples of 4 with zero weights
  }
  if (wave == 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) sm.split[j][lane] = twiddle[lane + 64 * j];      // split twiddle: theta = 2 pi k / 1024
  }
  if (wave == 2 && (lane & 31) == 0) {
    // bin 512 is the only live column of the 17th 32-bin block: its pseudo-inverse row as plain fp32 (h0 + h1 of the
    // pre-split table, unscaled), so that the matrix-core loop below runs 16 blocks -- two per wave, none with three
    const int hf = lane >> 5;
    const float un = pinv_unscale[512];
#pragma unroll
    for (int st = 0; st < kSteps; ++st) {
      const f16x8 h0 = __builtin_bit_cast(f16x8, pinv_pairs[(((kNBlocks - 1) * kSteps + st) * 2 + 0) * 64 + lane]);
      const f16x8 h1 = __builtin_bit_cast(f16x8, pinv_pairs[(((kNBlocks - 1) * kSteps + st) * 2 + 1) * 64 + lane]);
#pragma unroll
      for (int i = 0; i < 8; ++i) sm.p512[16 * st + 8 * hf + i] = ((float)h0[i] + (float)h1[i]) * un;
    }
  }
  __syncthreads();
  if (tid < kMels) {
    int off = 0;
    for (int m = 0; m < tid; ++m) off += sm.band[kMels + m];
    sm.band[2 * kMels + tid] = off * 4;
  }
  const bool aligned = ((nsamps | nhop) & 1) == 0;
  // The two workgroups of a CU (blockIdx and blockIdx + grid / 2 under the round-robin dispatch) start half a tile apart,
  // so that one's FFT (vector ALUs) runs next to the other's projection and stores (L2 / HBM); the stagger is stable
  // because the overlapped schedule is the faster one for both.
  if (skew_sleeps > 0 && blockIdx.x >= gridDim.x / 2)
    for (int i = 0; i < skew_sleeps; ++i) __builtin_amdgcn_s_sleep(127);

  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const int clip = tile / tiles_per_clip;
    const int f0 = (tile - clip * tiles_per_clip) * kTile;
    const int nrows = T - f0 < kTile ? T - f0 : kTile;
    const int64_t row0 = (int64_t)clip * T + f0;

    // ---- FFT of frames f0 + 2 wave, + 1 (zero samples beyond the clip, as tf.contrib.signal.stft pads).  The body is
    // stft1024_kernel's, inlined by hand: routed through fft1024.h's functions the register allocation needs ~22 more
    // registers, which at four waves per SIMD (two workgroups per CU) means 400-600 bytes of scratch per lane ----
    {
      const int fa = f0 + 2 * wave;
      const bool va = fa < T, vb = fa + 1 < T;
      int lf = lane;
      asm volatile("" : "+v"(lf));
      const int hi = lf >> 3, lo = lf & 7;
      if (va) {
        // hop 256 (the launcher refuses other hops): frames fa and fa + 1 share 768 samples -- ten loads per lane, and the
        // rest of stft1024_hop256_kernel's economies (stft.hip): 1/2 in the window table, ds_read_b64 transposes, both bins
        // of a split pair from the lane that owns the lower one
        f2 re[8], im[8];
        {
          const float* src = wav + (int64_t)clip * nsamps;
          const int64_t s0 = (int64_t)fa * 256;
          float2 raw[10];
          if (aligned && s0 + 256 + kNfft <= nsamps) {
            const float* p0 = src + s0;
#pragma unroll
            for (int j = 0; j < 10; ++j) raw[j] = *reinterpret_cast<const float2*>(p0 + (128u * j + 2u * (unsigned)lane));
          } else {
#pragma unroll
            for (int j = 0; j < 10; ++j) {
              const int64_t i0 = s0 + 128 * j + 2 * lane;
              raw[j].x = i0 < nsamps ? src[i0] : 0.f;
              raw[j].y = i0 + 1 < nsamps ? src[i0 + 1] : 0.f;
            }
          }
#pragma unroll
          for (int a = 0; a < 8; ++a) {
            const float2 w = sm.tables.win[a][lane];          // halved when the tables were filled
            re[a] = f2{raw[a].x * w.x, raw[a + 2].x * w.x};
            im[a] = f2{raw[a].y * w.y, raw[a + 2].y * w.y};
          }
        }
        f2* const wr = plane + hi * 9 + lo;                                  // + 72 p: element (8 p + hi) * 9 + lo
        const unsigned rd1 = advoc::lds_address(plane + 72 * hi + lo);       // + 72 b bytes
        const unsigned rd2 = advoc::lds_address(plane + 9 * lf);             // + 8 c bytes
        // pass 1: DFT over a -> p, twiddle, transpose (b,c | p) -> (p,c | b)
        dft8(re, im);
        {
          f2 ti[8];
#pragma unroll
          for (int p = 0; p < 8; ++p) {
            f2 r = re[p], i = im[p];
            if (p) {                      // W64^0 = 1
              const float2 t = sm.tables.t1[p][lane];
              r = re[p] * t.x - im[p] * t.y;
              i = re[p] * t.y + im[p] * t.x;
            }
            wr[72 * p] = r;
            ti[p] = i;
          }
          ADVOC_FFT_LDS_READ8(re, rd1, 0, 72);
          ADVOC_FFT_LDS_WAIT8(re);
#pragma unroll
          for (int p = 0; p < 8; ++p) wr[72 * p] = ti[p];
          ADVOC_FFT_LDS_READ8(im, rd1, 0, 72);
          ADVOC_FFT_LDS_WAIT8(im);
        }
        // pass 2: DFT over b -> q, twiddle, transpose (p,c | q) -> (q,p | c)
        dft8(re, im);
        {
          f2 ti[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float2 t = sm.tables.t2[q][lane];
            wr[72 * q] = re[q] * t.x - im[q] * t.y;
            ti[q] = re[q] * t.y + im[q] * t.x;
          }
          ADVOC_FFT_LDS_READ8(re, rd2, 0, 8);
          ADVOC_FFT_LDS_WAIT8(re);
#pragma unroll
          for (int q = 0; q < 8; ++q) wr[72 * q] = ti[q];
          ADVOC_FFT_LDS_READ8(im, rd2, 0, 8);
          ADVOC_FFT_LDS_WAIT8(im);
        }
        // pass 3: DFT over c -> r.  Lane now holds Z[lane + 64 r] / 2.
        dft8(re, im);
        // real-FFT split, two bins per step (X[k] = S - T, X[512 - k] = conj(S + T)), |.|: to HBM and into the wave's plane
        // (every read of the plane above has returned)
        const int partner = ((64 - lane) & 63) * 4;
        float* orow0 = mag_out + (row0 + 2 * wave) * kBins;
        float* orow1 = orow0 + kBins;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f2 c, d;
          c.x = bperm(partner, re[7 - r].x); c.y = bperm(partner, re[7 - r].y);
          d.x = bperm(partner, im[7 - r].x); d.y = bperm(partner, im[7 - r].y);
          if (lane == 0) {  // k = 64 r pairs with 64 (8 - r) on the same lane (Z[512] = Z[0])
            c = re[(8 - r) & 7];
            d = im[(8 - r) & 7];
          }
          const float2 cs = sm.split[r][lane];
          const f2 a = re[r], b = im[r];
          const f2 sr = a + c, si = b - d, dr = a - c, di = b + d;
          const f2 tr = cs.y * dr - cs.x * di;
          const f2 ti = cs.y * di + cs.x * dr;
          const f2 xkr = sr - tr, xki = si - ti, xmr = sr + tr, xmi = si + ti;
          const f2 k2 = xkr * xkr + xki * xki, m2 = xmr * xmr + xmi * xmi;
          const f2 mk = {__builtin_amdgcn_sqrtf(k2.x), vb ? __builtin_amdgcn_sqrtf(k2.y) : 0.f};
          const f2 mm = {__builtin_amdgcn_sqrtf(m2.x), vb ? __builtin_amdgcn_sqrtf(m2.y) : 0.f};
          const unsigned k = (unsigned)lane + 64u * r, m = 512u - k;
          orow0[k] = mk.x;
          orow0[m] = mm.x;
          if (vb) {
            orow1[k] = mk.y;
            orow1[m] = mm.y;
          }
          plane[k] = mk;
          plane[m] = mm;
        }
        if (lane == 0) {  // bin 256 pairs with itself: X[256] = conj(Z[256])
          const f2 xr = 2.0f * re[4], xi = 2.0f * im[4];
          const f2 m2 = xr * xr + xi * xi;
          const f2 mg = {__builtin_amdgcn_sqrtf(m2.x), vb ? __builtin_amdgcn_sqrtf(m2.y) : 0.f};
          orow0[256] = mg.x;
          if (vb) orow1[256] = mg.y;
          plane[256] = mg;
        } else {
          plane[512 + lane] = (f2){0.f, 0.f};      // zero slack behind the Nyquist bin for the padded filterbank runs
        }
      } else {
#pragma unroll
        for (int r = 0; r < 9; ++r) plane[lane + 64 * r] = (f2){0.f, 0.f};
      }
    }
    __syncthreads();

    // ---- mel = mag W^T on the bands' runs of bins: thread -> (band, frame) with the frame fastest ----
    // (per-thread constants of this phase and of the next are derived from OPAQUE copies of the thread id inside the tile
    // loop: hoisted out of it as loop invariants they stay alive across the FFT, which then spills)
    int tm = tid;
    asm volatile("" : "+v"(tm));
    constexpr int kOut = (kTile * kMels + kXThreads - 1) / kXThreads;
    float macc[kOut];
#pragma unroll
    for (int t = 0; t < kOut; ++t) {
      const int o = tm + t * kXThreads;
      float acc = 0.f;
      if (o < kTile * kMels) {
        const int mb = o >> 4, fr = o & 15;
        const int lo = sm.band[mb], n4 = sm.band[kMels + mb], wo = sm.band[2 * kMels + mb];
        const float* x = reinterpret_cast<const float*>(&sm.planes[fr >> 1][lo]) + (fr & 1);
        const float* w = sm.w + wo;
        for (int k = 0; k < n4; ++k) {
          const float4 w4 = *reinterpret_cast<const float4*>(w + 4 * k);
          acc = fmaf(x[8 * k], w4.x, acc); acc = fmaf(x[8 * k + 2], w4.y, acc);
          acc = fmaf(x[8 * k + 4], w4.z, acc); acc = fmaf(x[8 * k + 6], w4.w, acc);
        }
      }
      macc[t] = acc;
    }
#pragma unroll
    for (int t = 0; t < kOut; ++t) {
      const int o = tm + t * kXThreads;
      if (o < kTile * kMels) sm.mel[(o & 15) * kMelPitch + (o >> 4)] = macc[t];
    }
    __syncthreads();          // every read of the planes is done (the next FFT may overwrite them); the mel tile is complete
    {
      float* dst = mel_out + row0 * kMels;                // the tile's rows are contiguous
      for (int i = tm; i < nrows * kMels; i += kXThreads) dst[i] = sm.mel[(i / kMels) * kMelPitch + (i % kMels)];
    }
    int lp = lane;
    asm volatile("" : "+v"(lp));
    const int l32 = lp & 31, half = lp >> 5;

    // ---- inv = mel P^T: mel row l32 (< 16; rows 16..31 of the MFMA tile are zero) as an fp16 pair under its own scale ----
    f16x8 a0[kSteps], a1[kSteps];
    float inv_sa;
    {
      float v[kSteps][8];
      float amax = 0.f;
#pragma unroll
      for (int st = 0; st < kSteps; ++st)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          v[st][i] = l32 < kTile ? sm.mel[l32 * kMelPitch + 16 * st + 8 * half + i] : 0.f;
          amax = fmaxf(amax, fabsf(v[st][i]));
        }
      amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
      const float up = pair_scale(amax);
      inv_sa = 1.f / up;
#pragma unroll
      for (int st = 0; st < kSteps; ++st) pair_split8(v[st], up, a0[st], a1[st]);
    }
    // accumulator register r (< 8) of this lane is frame (r & 3) + 8 (r >> 2) + 4 half
    float unrow[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) unrow[r] = __shfl(inv_sa, (r & 3) + 8 * (r >> 2) + 4 * half, 64);
    if (wave == 7) {
      // bin 512 of the tile's frames on the vector ALUs: lane = (frame, quarter of the 80 mel bands), two xor steps
      const int f = lp & 15, qd = lp >> 4;
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < kMels / 4; ++i) a = fmaf(sm.mel[f * kMelPitch + qd * (kMels / 4) + i], sm.p512[qd * (kMels / 4) + i], a);
      a += __shfl_xor(a, 16, 64);
      a += __shfl_xor(a, 32, 64);
      if (qd == 0 && f < nrows) inv_out[(row0 + f) * kBins + 512] = a;
    }
#pragma unroll
    for (int q = 0; q < (kNBlocks - 1) / kXWaves; ++q) {
      const int nb = wave + q * kXWaves;
      const int n = nb * 32 + l32;
      const float unsb = pinv_unscale[n];
      floatx16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      // (the operand of one k step at a time: ten fragments held at once tipped the kernel into scratch at 128 registers)
#pragma unroll
      for (int st = 0; st < kSteps; ++st) {
        const f16x8 b0 = __builtin_bit_cast(f16x8, pinv_pairs[((nb * kSteps + st) * 2 + 0) * 64 + lp]);
        const f16x8 b1 = __builtin_bit_cast(f16x8, pinv_pairs[((nb * kSteps + st) * 2 + 1) * 64 + lp]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[st], b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[st], b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[st], b0, acc, 0, 0, 0);
      }
      if (n < kBins) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int f = (r & 3) + 8 * (r >> 2) + 4 * half;
          if (f < nrows) inv_out[(row0 + f) * kBins + n] = acc[r] * (unrow[r] * unsb);
        }
      }
    }
  }
}

}  // namespace

extern "C" int advoc_stft_mel_pinv_f32(const float* wav, int64_t batch, int64_t nsamps, const float* window,
                                       const float* twiddle, int32_t nfft, int32_t nhop, int64_t nframes,
                                       const float* mel_wp, const int32_t* band_lo_hi, int32_t packed_weights, int32_t bins,
                                       int32_t n_mels, const void* pinv_pairs, const float* pinv_unscale, float* mag,
                                       float* mel, float* inv, advoc_stream_t stream) {
  if (batch < 0 || nsamps < 0 || nframes < 0 || nhop <= 0) return ADVOC_ERR_BAD_SHAPE;
  if (nfft != kNfft || nhop != 256 || bins != kBins || n_mels != kMels || packed_weights < 0 ||
      packed_weights > kMaxW || packed_weights % 4)
    return ADVOC_ERR_UNSUPPORTED;                      // callers fall back to advoc_stft_mag_f32 + advoc_mel_pinv_f32
  if (batch == 0 || nframes == 0) return ADVOC_OK;
  if (!wav || !window || !twiddle || !mel_wp || !band_lo_hi || !pinv_pairs || !pinv_unscale || !mag || !mel || !inv)
    return ADVOC_ERR_NULL;
  if ((reinterpret_cast<uintptr_t>(wav) & 7) || (reinterpret_cast<uintptr_t>(pinv_pairs) & 15)) return ADVOC_ERR_UNSUPPORTED;
  if (nframes > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
  const int64_t tiles_per_clip = advoc::ceil_div(nframes, kTile);
  const int64_t total = batch * tiles_per_clip;
  if (total > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
  constexpr int lds = (int)sizeof(XSmem);
  static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
  static const int minw = [] { const char* e = getenv("ADVOC_EXTRACT_WAVES"); return e && atoi(e) == 2 ? 2 : 4; }();      // A/B switch: workgroups per CU = minw / 2
  auto kern = minw == 4 ? stft_mel_pinv_kernel<4> : stft_mel_pinv_kernel<2>;
  const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (attr != hipSuccess) { advoc::note_hip_error(attr); return ADVOC_ERR_HIP; }
  static const int resident = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    int n = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    (void)hipGetLastError();
    return n > 0 ? n : 256;
  }();
  constexpr int skew_env = 0;      // (start stagger of the CU's second workgroup: measured, no change -- the switch is gone, r6)
  const int skew = minw == 4 ? skew_env : 0;
  const int64_t per_cu = minw == 4 ? 2 : 1;
  const int64_t grid = total < per_cu * resident ? total : per_cu * (int64_t)resident;      // persistent
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kXThreads), lds, advoc::as_stream(stream), wav, nsamps,
                     window, reinterpret_cast<const float2*>(twiddle), nhop, (int)nframes, mel_wp,
                     reinterpret_cast<const int2*>(band_lo_hi), (int)packed_weights,
                     reinterpret_cast<const uint4*>(pinv_pairs), pinv_unscale, mag, mel, inv, (int)tiles_per_clip, (int)total, skew);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}
