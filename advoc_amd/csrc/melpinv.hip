// Mel projection and its pseudo-inverse in ONE pass over the magnitudes: mag [rows, 513] -> mel [rows, 80] = mag W^T and
// inv [rows, 513] = mel P^T (W the mel filterbank [80, 513], P its pseudo-inverse [513, 80]).  Replaces the two
// tf.tensordot / tf.matmul launches of models/advoc/spectral_util.py:29-43 as models/advoc/train_evaluate.py:55-56 chains
// them for every training batch (the "x_inverted" input of the generator).
//
// The two advoc_matmul_nt_f32 launches it replaces run an LDS-tiled FMA kernel (34 TFLOP/s) and read the 2 KiB rows
// twice.  Here one PERSISTENT workgroup per CU walks tiles of 32 frames (band_lo_hi must describe at most 2 048 weights
// in total: the caller checks):
//   * the tile's 32 x 513 magnitudes go to LDS once, by LDS-DMA (66 KB); the next tile's are fetched under the current
//     tile's MFMAs;
//   * the filterbank is triangular -- every band is a short run of bins -- so the mel projection is ~1 000 FMAs per
//     frame on the vector ALUs; the caller passes the runs' weights PACKED back to back (each run padded with zeros to
//     a multiple of 4), a few KB that live in LDS for the whole launch;
//   * the 32 x 80 mel tile (in LDS, over the magnitudes it came from) is the A operand of the pseudo-inverse on the
//     f16 matrix cores with igemm_h3.hip's arithmetic: every mel ROW under its own power-of-two scale as an fp16 pair,
//     every pseudo-inverse COLUMN likewise, three products per 32x32x16 step (a0 b1 + a1 b0 + a0 b0), fp32
//     accumulation, exact unscaling per row and column -- 15 MFMAs of 32 cycles per 32 x 32 output block (17 blocks
//     over 8 waves) where v_mfma_f32_32x32x2_f32 takes 40 of 64; error against float64 below 5e-6 of the row's
//     largest output for rows from 1e-6 to 1e3 in one tile (tests/test_hip_spectral.py).  The pseudo-inverse is the STATIONARY operand: every wave keeps the columns of its 2-3 output blocks in registers for
//     the whole launch (P arrives transposed, [80][513], so that this one-time load reads 128 contiguous bytes per
//     k-step; re-fetched per tile it was 680 four-byte loads per tile, more vector-memory instructions than everything
//     else together);
//   * algorithmic HBM bytes per frame: 2 052 read + 320 + 2 052 written.
// Measured for 131 072 frames (tools/micro/melpinv_time.py): 168 us (3.5 TB/s algorithmic) against 650-680 us for the
// two projection launches; steps on the way: per-lane (frame, band) with the band fastest 580 us, load -> ds_write loop
// 560, DMA + packed weights 328, 8 waves + prefetched operand 271, persistent + stationary operand 223 (fp32 MFMA: ~90 us
// of matrix-pipe time per launch), fp16 pairs 168 us = 3.5 TB/s algorithmic.  The floor is ~100 us of HBM time.
#include <hip/hip_fp16.h>

#include "common.h"

#ifndef ADVOC_MELPINV_WAVES
#define ADVOC_MELPINV_WAVES 8
#endif

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_void_p;

constexpr int kBins = 513, kMels = 80, kFrames = 32;
constexpr int kMagPitch = 513;         // floats per LDS row of magnitudes: the tile is a linear copy (LDS-DMA)
constexpr int kMelPitch = 81;          // floats per LDS row of the mel tile (odd: the A-fragment reads walk rows)
constexpr int kNBlocks = (kBins + 31) / 32;   // 17 output blocks of 32 bins
constexpr int kWaves = ADVOC_MELPINV_WAVES, kThreads = 64 * kWaves;   // one persistent workgroup per CU
constexpr int kMagFloats = (kFrames * kBins + 63) / 64 * 64;   // the DMA writes whole 64-lane blocks
constexpr int kMaxW = 2048;            // packed filterbank weights the kernel takes (the reference's bank has ~1 100)

// power of two that takes `amax` into [2^13, 2^14) (1 for zero / non-finite): x * up = h0 + h1 in fp16 to 2^-22 of amax
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

__global__ __launch_bounds__(kThreads, 1) void mel_pinv_kernel(const float* __restrict__ mag, const float* __restrict__ mel_wp,
                                                          const int2* __restrict__ band, const float* __restrict__ inv_wt,
                                                          float* __restrict__ mel_out, float* __restrict__ inv_out,
                                                          int64_t rows, int packed, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_mag = sm;                                   // [32][513] + slack to a whole DMA block
  float* s_mel = sm;                                   // [32][81], over the magnitudes once the mel phase has read them
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* s_w = sm + kMagFloats;                        // the filterbank's runs of weights, packed (<= kMaxW floats)
  int* s_band = reinterpret_cast<int*>(s_w + kMaxW);   // [3][80]: first bin, run length / 4, offset into s_w
  float* s_p512 = reinterpret_cast<float*>(s_band + 3 * kMels);      // [80]: the pseudo-inverse row of bin 512 (fp32)
  const int l32 = lane & 31, half = lane >> 5;

  const int64_t bytes = rows * kBins * 4;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(mag), 0, bytes > 0xffffffffLL ? (int)0xffffffff : (int)bytes, 0x00020000);
  // Magnitudes of tile T -> LDS by LDS-DMA (buffer_load_dword ... lds: 256 contiguous bytes per instruction, no
  // registers, every instruction of a thread in flight at once; a plain load -> ds_write loop waits one memory latency
  // per iteration: 0.56 ms for 131 072 frames).  The tile is a linear copy, row pitch 513 floats: the 32 frames of a
  // band step fall on 32 different banks.  Rows past the end of the tensor are beyond num_records and arrive as zeros,
  // and so does the slack behind the tile (the padded runs may read up to 3 floats of it).
#define ADVOC_MP_LOAD(T)                                                                                 \
  {                                                                                                      \
    const unsigned base_ = (unsigned)((int64_t)(T) * kFrames * kBins * 4);       /* < 2^32 (launcher) */  \
    for (int c = wave; c < kMagFloats / 64; c += kWaves) {                                               \
      const bool in_tile_ = c * 64 + lane < kFrames * kBins;                                             \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_p)(s_mag + c * 64), 4,                      \
                                               in_tile_ ? (int)(base_ + (unsigned)(c * 64 + lane) * 4u)  \
                                                        : (int)0xffffffff, 0, 0, 0);                     \
    }                                                                                                    \
  }

  // ---- once per workgroup (it walks tiles): the packed filterbank runs -> LDS, the pseudo-inverse columns of this
  // wave's output blocks -> REGISTERS (the stationary operand: re-fetching them per tile was 680 four-byte loads per
  // tile, more vector-memory instructions than everything else together) ----
  int tile = blockIdx.x;
  if (tile < ntiles) ADVOC_MP_LOAD(tile);
  {
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(mel_wp), 0, packed * 4, 0x00020000);
    for (int c = wave; c * 64 < packed; c += kWaves)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_p)(s_w + c * 64), 4, (c * 64 + lane) * 4, 0, 0, 0);
  }
  if (tid < kMels) {
    const int2 bnd = band[tid];
    s_band[tid] = bnd.x;
    s_band[kMels + tid] = (bnd.y - bnd.x + 3) >> 2;     // runs are padded to multiples of 4 with zero weights
    s_p512[tid] = inv_wt[tid * kBins + (kBins - 1)];
  }
  // (r4) bin 512 is the only live column of the 17th block: it is computed on the vector ALUs by the last wave (32 frames x
  // 80 bands: 40 multiply-adds per lane), so that the matrix-core loop has 16 blocks -- two per wave, none with three (the
  // wave with three set the length of the phase) -- and 40 registers less of stationary operand
  constexpr int kMfmaBlocks = kNBlocks - 1;
  static_assert(kMfmaBlocks * 32 + 1 == kBins, "the last block holds one bin");
  constexpr int kBPW = (kMfmaBlocks + kWaves - 1) / kWaves;   // output blocks per wave: 2
  constexpr int kSteps = kMels / 16;                   // 5 MFMA k-steps of 16 mel bands
  // lane (l32, half) of a 32x32x16 MFMA holds k = 16 s + 8 half + 0..7 of row / column l32.  The pseudo-inverse column
  // of output bin n as an fp16 PAIR under its own power-of-two scale (igemm_h3.hip's arithmetic: x 2^s = h0 + h1 to
  // 2^-22, three products a0 b1 + a1 b0 + a0 b0, fp32 accumulation, exact unscaling): 15 MFMAs of 32 cycles per output
  // block instead of 40 fp32 MFMAs of 64
  f16x8 b0[kBPW][kSteps], b1[kBPW][kSteps];
  float inv_sb[kBPW];
#pragma unroll
  for (int q = 0; q < kBPW; ++q) {
    const int n = (wave + q * kWaves) * 32 + l32;
    const bool live = wave + q * kWaves < kMfmaBlocks;
    const float* pcol = inv_wt + (n < kBins ? n : kBins - 1);
    float v[kSteps][8];
    float amax = 0.f;
#pragma unroll
    for (int st = 0; st < kSteps; ++st)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v[st][i] = live ? pcol[(16 * st + 8 * half + i) * kBins] : 0.f;
        amax = fmaxf(amax, fabsf(v[st][i]));
      }
    amax = fmaxf(amax, __shfl_xor(amax, 32, 64));       // the other half of the column's 80 values
    const float up = pair_scale(amax);
    inv_sb[q] = 1.f / up;
#pragma unroll
    for (int st = 0; st < kSteps; ++st) pair_split8(v[st], up, b0[q][st], b1[q][st]);
  }
  __syncthreads();
  if (tid < kMels) {                                   // offset of run m in the packed array (independent LDS reads)
    int off = 0;
    for (int m = 0; m < tid; ++m) off += s_band[kMels + m];
    s_band[2 * kMels + tid] = off * 4;
  }

  for (; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = (int64_t)tile * kFrames;
    const int nrows = rows - r0 < kFrames ? (int)(rows - r0) : kFrames;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- mel = mag W^T on the bands' runs of bins.  Thread -> (band, frame) with the FRAME fastest: the 32 lanes of a
    // half wave share one band, so its weights are an LDS broadcast and the run length is uniform ----
    constexpr int kOutPerThread = (kFrames * kMels + kThreads - 1) / kThreads;
    float macc[kOutPerThread];
#pragma unroll
    for (int t = 0; t < kOutPerThread; ++t) {
      const int o = tid + t * kThreads;
      float acc = 0.f;
      if (o < kFrames * kMels) {
        const int m = o >> 5, f = o & 31;
        const int lo = s_band[m], n4 = s_band[kMels + m], wo = s_band[2 * kMels + m];
        const float* x = s_mag + f * kMagPitch + lo;
        const float* w = s_w + wo;
        for (int k = 0; k < n4; ++k) {
          const float x0 = x[4 * k], x1 = x[4 * k + 1], x2 = x[4 * k + 2], x3 = x[4 * k + 3];
          const float4 w4 = *reinterpret_cast<const float4*>(w + 4 * k);
          acc = fmaf(x0, w4.x, acc); acc = fmaf(x1, w4.y, acc); acc = fmaf(x2, w4.z, acc); acc = fmaf(x3, w4.w, acc);
        }
      }
      macc[t] = acc;
    }
    __syncthreads();                                    // every read of the magnitudes is done: the mel tile takes their place
#pragma unroll
    for (int t = 0; t < kOutPerThread; ++t) {
      const int o = tid + t * kThreads;
      if (o < kFrames * kMels) s_mel[(o & 31) * kMelPitch + (o >> 5)] = macc[t];
    }
    __syncthreads();
    {
      float* dst = mel_out + r0 * kMels;                // the tile's 32 x 80 outputs are contiguous
      for (int i = tid; i < nrows * kMels; i += kThreads) dst[i] = s_mel[(i / kMels) * kMelPitch + (i % kMels)];
    }

    // ---- inv = mel P^T on the f16 matrix cores: the mel row of frame l32 as an fp16 pair under the ROW's own scale ----
    f16x8 a0[kSteps], a1[kSteps];
    float inv_sa;
    {
      float v[kSteps][8];
      float amax = 0.f;
#pragma unroll
      for (int st = 0; st < kSteps; ++st)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          v[st][i] = s_mel[l32 * kMelPitch + 16 * st + 8 * half + i];
          amax = fmaxf(amax, fabsf(v[st][i]));
        }
      amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
      const float up = pair_scale(amax);
      inv_sa = 1.f / up;
#pragma unroll
      for (int st = 0; st < kSteps; ++st) pair_split8(v[st], up, a0[st], a1[st]);
    }
    // accumulator register r of this lane is frame f(r) = (r & 3) + 8 (r >> 2) + 4 half: its row's unscale factor
    float unrow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) unrow[r] = __shfl(inv_sa, (r & 3) + 8 * (r >> 2) + 4 * half, 64);
    float v512 = 0.f;
    if (wave == kWaves - 1) {                           // bin 512: frame l32, bands [40 half, 40 half + 40)
#pragma unroll
      for (int i = 0; i < kMels / 2; ++i) v512 = fmaf(s_mel[l32 * kMelPitch + half * (kMels / 2) + i], s_p512[half * (kMels / 2) + i], v512);
      v512 += __shfl_xor(v512, 32, 64);
    }
    __syncthreads();                                    // the mel tile is in registers: the next tile's magnitudes may land
    if (tile + (int)gridDim.x < ntiles) ADVOC_MP_LOAD(tile + (int)gridDim.x);
    if (wave == kWaves - 1 && half == 0 && l32 < nrows) inv_out[(r0 + l32) * kBins + (kBins - 1)] = v512;
#pragma unroll
    for (int q = 0; q < kBPW; ++q) {
      const int nb = wave + q * kWaves;
      if (nb >= kMfmaBlocks) break;
      floatx16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int st = 0; st < kSteps; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[st], b1[q][st], acc, 0, 0, 0);
#pragma unroll
      for (int st = 0; st < kSteps; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[st], b0[q][st], acc, 0, 0, 0);
#pragma unroll
      for (int st = 0; st < kSteps; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[st], b0[q][st], acc, 0, 0, 0);
      const int n = nb * 32 + l32;
      if (n < kBins) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int f = (r & 3) + 8 * (r >> 2) + 4 * half;
          if (f < nrows) inv_out[(r0 + f) * kBins + n] = acc[r] * (unrow[r] * inv_sb[q]);
        }
      }
    }
  }
#undef ADVOC_MP_LOAD
}

}  // namespace

extern "C" int advoc_mel_pinv_f32(const float* mag, const float* mel_wp, const int32_t* band_lo_hi, const float* inv_wt,
                                  float* mel_out, float* inv_out, int64_t rows, int32_t bins, int32_t n_mels,
                                  int32_t packed_weights, advoc_stream_t stream) {
  if (!mag || !mel_wp || !band_lo_hi || !inv_wt || !mel_out || !inv_out) return ADVOC_ERR_NULL;
  if (rows < 0) return ADVOC_ERR_BAD_SHAPE;
  if (bins != kBins || n_mels != kMels || packed_weights < 0 || packed_weights > kMaxW || packed_weights % 4)
    return ADVOC_ERR_UNSUPPORTED;                                          // callers fall back to advoc_matmul_nt_f32
  if (rows == 0) return ADVOC_OK;
  const int64_t blocks = advoc::ceil_div(rows, kFrames);
  if (blocks > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
  constexpr int lds = (kMagFloats + kMaxW + 4 * kMels) * (int)sizeof(float);
  static_assert(kFrames * kMelPitch <= kMagFloats && lds <= 160 * 1024, "LDS budget");
  if (rows * kBins * 4 > 0xffffffffLL) return ADVOC_ERR_UNSUPPORTED;         // 32-bit buffer offsets
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(mel_pinv_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (attr != hipSuccess) { advoc::note_hip_error(attr); return ADVOC_ERR_HIP; }
  ADVOC_CLEAR_LAUNCH_ERROR();
  // persistent workgroups: one per CU (the stationary operand takes the registers), each walks tiles (blockIdx,
  // blockIdx + grid, ...) and fetches the next tile's magnitudes under the current tile's MFMAs
  static const int resident = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    int n = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    (void)hipGetLastError();
    return n > 0 ? n : 256;
  }();
  const int64_t grid = blocks < resident ? blocks : resident;
  hipLaunchKernelGGL(mel_pinv_kernel, dim3((unsigned)grid), dim3(kThreads), lds, advoc::as_stream(stream), mag, mel_wp,
                     reinterpret_cast<const int2*>(band_lo_hi), inv_wt, mel_out, inv_out, rows, (int)packed_weights,
                     (int)blocks);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}
