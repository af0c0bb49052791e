// STFT-magnitude feature extractor for gfx950 (MI355X).
//
// Replaces tf.contrib.signal.stft (+ tf.abs) reached from the reference at
// advoc/spectral.py:60-83 and advoc/loader.py:116-128: frame 1024 / hop 256, multiply by the
// lws sqrt-Hann window, real FFT, |.|.
//
// Mapping to the hardware
//   * one workgroup (4 wavefronts) = 16 consecutive frames of one clip.  The
//     (16-1)*hop+1024 sample span is read from HBM once, coalesced, into LDS: the 75 %
//     overlap between neighbouring frames is served from LDS, not re-fetched.
//   * one wavefront = one frame at a time.  The 1024-point real FFT is a 512-point complex
//     FFT (z[n] = x[2n] + i x[2n+1]) held 8 complex values per lane, computed as three
//     radix-8 passes (512 = 8*8*8); the two inter-pass transposes go through a private,
//     bank-conflict-free LDS plane per wave (no workgroup barrier inside the frame loop).
//   * twiddles and the window live in registers for the life of the kernel (they depend only
//     on the lane), evaluated once with sincospif.
//   * the real-FFT split pairs Z[k] with Z[512-k] by one cross-lane permute per value, then
//     each lane stores 8 magnitudes: 256 B contiguous per store instruction.
// HBM-bound: algorithmic traffic is 256 new samples in + 513 floats out per frame.
#include <math.h>

#include "common.h"

namespace {

using advoc::wave_lds_sync;

constexpr int kNfft = 1024;
constexpr int kBins = kNfft / 2 + 1;
constexpr int kWaves = 4;
constexpr int kFramesPerWave = 2;
constexpr int kFramesPerBlock = kWaves * kFramesPerWave;
constexpr int kPlane = 576;  // 64 rows x 9 floats (8 + 1 pad): conflict-free transposes

// forward 8-point DFT in place: v[p] = sum_a v[a] * exp(-2*pi*i*a*p/8)
__device__ __forceinline__ void dft8(float (&re)[8], float (&im)[8]) {
  const float h = 0.70710678118654752440f;
  // radix-2 DIF stage: sums feed even outputs, twiddled differences feed odd outputs
  float sr[4], si[4], dr[4], di[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    sr[n] = re[n] + re[n + 4];
    si[n] = im[n] + im[n + 4];
    dr[n] = re[n] - re[n + 4];
    di[n] = im[n] - im[n + 4];
  }
  // d[n] *= W8^n
  {
    float r1 = (dr[1] + di[1]) * h, i1 = (di[1] - dr[1]) * h;  // * (1 - i)/sqrt2
    dr[1] = r1; di[1] = i1;
    float r2 = di[2], i2 = -dr[2];                             // * (-i)
    dr[2] = r2; di[2] = i2;
    float r3 = (di[3] - dr[3]) * h, i3 = -(dr[3] + di[3]) * h; // * (-1 - i)/sqrt2
    dr[3] = r3; di[3] = i3;
  }
  // 4-point DFT of s -> even bins, of d -> odd bins
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    float* xr = half ? dr : sr;
    float* xi = half ? di : si;
    float b0r = xr[0] + xr[2], b0i = xi[0] + xi[2];
    float b2r = xr[0] - xr[2], b2i = xi[0] - xi[2];
    float b1r = xr[1] + xr[3], b1i = xi[1] + xi[3];
    float b3r = xi[1] - xi[3], b3i = -(xr[1] - xr[3]);  // (x1 - x3) * (-i)
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
    int tiles_per_clip) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int span = (kFramesPerBlock - 1) * nhop + kNfft;
  float* stage = smem;
  const int span_pad = (span + 3) & ~3;

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t clip = blockIdx.x / tiles_per_clip;
  const int tile = blockIdx.x % tiles_per_clip;
  const int64_t f0 = (int64_t)tile * kFramesPerBlock;

  float* xr_plane = smem + span_pad + wave * (2 * kPlane);
  float* xi_plane = xr_plane + kPlane;

  // ---- stage the sample span (zero beyond the end of the clip: pad_end) ----
  {
    const float* src = wav + clip * nsamps;
    const int64_t s0 = f0 * nhop;
    if (((nsamps | nhop) & 3) == 0 && s0 + span <= nsamps) {   // interior tile, 16-byte aligned
      for (int i = 4 * threadIdx.x; i < span; i += 4 * kWaves * 64)
        *reinterpret_cast<float4*>(stage + i) = *reinterpret_cast<const float4*>(src + s0 + i);
    } else {
      for (int i = threadIdx.x; i < span; i += kWaves * 64) {
        const int64_t s = s0 + i;
        stage[i] = (s < nsamps) ? src[s] : 0.f;
      }
    }
  }

  // ---- per-lane constants ----
  const int hi = lane >> 3, lo = lane & 7;
  float w0[8], w1[8];
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    const float2 w = *reinterpret_cast<const float2*>(window + 128 * a + 2 * lane);
    w0[a] = w.x;
    w1[a] = w.y;
  }
  // twiddles: cos/sin(2 pi e / 1024) from the 1024-entry table built once on the host in double
  // precision (float2 {cos, sin} per entry, L2 resident): W_N^e = tw[e * 1024 / N] conjugated
  float t1r[8], t1i[8], t2r[8], t2i[8], tsn[8], tcs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    // pass-1 twiddle W64^(b*p): lane = (b=hi, c=lo), p = j
    const float2 a = twiddle[((hi * j) & 63) * 16];
    t1r[j] = a.x; t1i[j] = -a.y;
    // pass-2 twiddle W512^(c*(p+8q)): lane = (p=hi, c=lo), q = j
    const float2 b = twiddle[((lo * (hi + 8 * j)) & 511) * 2];
    t2r[j] = b.x; t2i[j] = -b.y;
    // split twiddle: theta = 2*pi*k/1024, k = lane + 64 j  (k <= 511)
    const float2 c = twiddle[lane + 64 * j];
    tcs[j] = c.x; tsn[j] = c.y;
  }
  const int partner = (64 - lane) & 63;

  __syncthreads();

  for (int fi = 0; fi < kFramesPerWave; ++fi) {
    const int fl = wave * kFramesPerWave + fi;
    const int64_t f = f0 + fl;
    if (f >= nframes) break;  // wave-uniform

    float re[8], im[8];
    // z[n] = x[2n] w[2n] + i x[2n+1] w[2n+1], n = 64 a + lane
    const float* fs = stage + fl * nhop;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const float2 v = *reinterpret_cast<const float2*>(fs + 128 * a + 2 * lane);
      re[a] = v.x * w0[a];
      im[a] = v.y * w1[a];
    }

    // pass 1: DFT over a -> p, twiddle, transpose (b,c | p) -> (p,c | b)
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

    // pass 2: DFT over b -> q, twiddle, transpose (p,c | q) -> (q,p | c)
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

    // pass 3: DFT over c -> r.  Lane now holds Z[lane + 64 r].
    dft8(re, im);

    // real-FFT split: X[k] = ((Zk + conj(Zm)) - i W1024^k (Zk - conj(Zm))) / 2, m = 512 - k
    float* orow = out + ((clip * nframes + f) * kBins) * (kComplexOut ? 2 : 1);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float c = __shfl(re[7 - r], partner, 64);
      float d = __shfl(im[7 - r], partner, 64);
      if (lane == 0) {  // k = 64 r pairs with 512 - 64 r = 64 (8 - r) on the same lane
        c = re[(8 - r) & 7];
        d = im[(8 - r) & 7];
      }
      const float a = re[r], b = im[r];
      const float sr = a + c, si = b - d, dr = a - c, di = b + d;
      const float xr = 0.5f * (sr - (tsn[r] * dr - tcs[r] * di));
      const float xi = 0.5f * (si - (tsn[r] * di + tcs[r] * dr));
      const int k = lane + 64 * r;
      if (kComplexOut) {
        *reinterpret_cast<float2*>(orow + 2 * k) = make_float2(xr, xi);
      } else {
        orow[k] = __builtin_amdgcn_sqrtf(xr * xr + xi * xi);   // v_sqrt_f32, <= 1 ulp
      }
    }
    if (lane == 0) {  // Nyquist bin: X[512] = Re Z0 - Im Z0
      const float xn = re[0] - im[0];
      if (kComplexOut) {
        *reinterpret_cast<float2*>(orow + 2 * (kBins - 1)) = make_float2(xn, 0.f);
      } else {
        orow[kBins - 1] = fabsf(xn);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Inverse: one wavefront = one frame.  irfft(1024) runs as the SAME 512-point complex transform
// (IDFT(Z) = conj(DFT(conj(Z))) / 512) after undoing the real-FFT split:
//   Ze[k] = (X[k] + conj(X[512-k])) / 2,  Zo[k] = (X[k] - conj(X[512-k])) / 2 * W1024^-k,
//   Z[k] = Ze[k] + i Zo[k];   x[2n] + i x[2n+1] = IDFT512(Z)[n].
// The frame is multiplied by the synthesis window and written to `frames` ([batch, T, 1024]);
// ola_kernel then sums the <= nfft/hop frames that cover each output sample (a gather: no atomics).
// Replaces lws.istft reached from advoc/spectral.py:300-309,320-321.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWaves * 64) void istft1024_frames_kernel(
    const float2* __restrict__ spec, const float* __restrict__ window, const float2* __restrict__ twiddle,
    int64_t total_frames, float* __restrict__ frames) {
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
  if (reinterpret_cast<uintptr_t>(wav) & 15) return ADVOC_ERR_UNSUPPORTED;   // float4 staging
  const int tiles = (int)advoc::ceil_div(nframes, kFramesPerBlock);
  const int64_t blocks = batch * tiles;
  if (blocks > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
  const int span = (kFramesPerBlock - 1) * nhop + kNfft;
  const size_t lds = sizeof(float) * (((span + 3) & ~3) + kWaves * 2 * kPlane);
  if (lds > 160 * 1024) return ADVOC_ERR_UNSUPPORTED;
  if (complex_out) {
    ADVOC_CLEAR_LAUNCH_ERROR();
    hipLaunchKernelGGL(stft1024_kernel<true>, dim3((unsigned)blocks), dim3(kWaves * 64), lds, stream,
                       wav, nsamps, window, reinterpret_cast<const float2*>(twiddle), nhop, nframes, out, tiles);
  } else {
    ADVOC_CLEAR_LAUNCH_ERROR();
    hipLaunchKernelGGL(stft1024_kernel<false>, dim3((unsigned)blocks), dim3(kWaves * 64), lds, stream,
                       wav, nsamps, window, reinterpret_cast<const float2*>(twiddle), nhop, nframes, out, tiles);
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

extern "C" int advoc_istft_f32(const float* spec, int64_t batch, int64_t nframes, const float* window,
                               const float* twiddle, int32_t nfft, int32_t nhop, float* frames_work,
                               float* wav, advoc_stream_t stream_) {
  hipStream_t stream = advoc::as_stream(stream_);
  if (batch < 0 || nframes < 0 || nhop <= 0) return ADVOC_ERR_BAD_SHAPE;
  if (nfft != kNfft || (nhop & 3) || nhop > kNfft) return ADVOC_ERR_UNSUPPORTED;
  if (batch == 0 || nframes == 0) return ADVOC_OK;
  if (!spec || !window || !twiddle || !frames_work || !wav) return ADVOC_ERR_NULL;
  const int64_t total = batch * nframes;
  int64_t blocks = advoc::ceil_div(total, kWaves);
  if (blocks > 4096) blocks = 4096;
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(istft1024_frames_kernel, dim3((unsigned)blocks), dim3(kWaves * 64), 0, stream,
                     reinterpret_cast<const float2*>(spec), window, reinterpret_cast<const float2*>(twiddle),
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
