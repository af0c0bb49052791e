// Mel projection kernels of the feature extractor (gfx950).
//
// advoc_matmul_nt_f32 replaces the TF MatMul/Tensordot ops at
// models/advoc/spectral_util.py:29-32 (mag[.,513] x W^T -> mel[.,80]),
// :34-43 (mel[.,80] x pinv(W)^T -> mag[.,513]) and advoc/spectral.py:204-208.
// advoc_mel_dbnorm_f32 replaces advoc/spectral.py:210-225.
//
// Both GEMMs are tiny (21 MFLOP per 256-frame clip) with K or N = 513 (not a multiple of
// anything): a plain LDS-tiled fp32 FMA kernel, bounds-checked on every edge.  Summation
// order over k is sequential ascending like a textbook dot product.
#include "common.h"

namespace {

constexpr int kTM = 64, kTN = 64, kTK = 16;

__global__ __launch_bounds__(256) void matmul_nt_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ w,
                                                        float* __restrict__ out, int64_t rows,
                                                        int K, int N) {
  __shared__ float xs[kTK][kTM + 1];
  __shared__ float ws[kTK][kTN + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * kTM;
  const int n0 = blockIdx.y * kTN;
  float acc[4][4] = {};
  // loader mapping: 256 threads cover a 64 x 16 tile, 4 elements each, k fastest
  const int lk = threadIdx.x & 15, lr = threadIdx.x >> 4;
  for (int k0 = 0; k0 < K; k0 += kTK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = lr + 16 * i;
      const int64_t r = r0 + rr;
      const int k = k0 + lk;
      xs[lk][rr] = (r < rows && k < K) ? x[r * K + k] : 0.f;
      const int n = n0 + rr;
      ws[lk][rr] = (n < N && k < K) ? w[(int64_t)n * K + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < kTK; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = xs[kk][ty + 16 * i];
        b[i] = ws[kk][tx + 16 * i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t r = r0 + ty + 16 * i;
    if (r >= rows) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx + 16 * j;
      if (n < N) out[r * N + n] = acc[i][j];
    }
  }
}

__global__ __launch_bounds__(256) void mel_dbnorm_kernel(float* __restrict__ v, int64_t count,
                                                         float min_level, float ref_db,
                                                         float min_db) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float inv_ln10 = 1.0f / logf(10.0f);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
    // 20 * (log(x) / log(10)) - ref  (advoc/spectral.py:213-218), then clip((db - min)/-min, 0, 1)
    const float db = 20.f * (logf(fmaxf(min_level, v[i])) * inv_ln10) - ref_db;
    const float n = (db - min_db) / -min_db;
    v[i] = fminf(fmaxf(n, 0.f), 1.f);
  }
}

// y = tanh(x) * scale + shift: the MelspecGAN generator's output non-linearity fused with
// feats_denorm (models/melspecgan/conv2d.py:139, util.py:11-12)
__global__ __launch_bounds__(256) void tanh_affine_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          int64_t n, float scale, float shift) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = tanhf(x[i]) * scale + shift;
}

}  // namespace

extern "C" int advoc_tanh_affine_f32(const float* x, float* y, int64_t n, float scale, float shift,
                                     advoc_stream_t stream) {
  if (n < 0) return ADVOC_ERR_BAD_SHAPE;
  if (n == 0) return ADVOC_OK;
  if (!x || !y) return ADVOC_ERR_NULL;
  const int64_t blocks = advoc::ceil_div(n, 256);
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(tanh_affine_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0,
                     advoc::as_stream(stream), x, y, n, scale, shift);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

extern "C" int advoc_matmul_nt_f32(const float* x, const float* w, float* out, int64_t rows,
                                   int32_t k, int32_t n, advoc_stream_t stream) {
  if (!x || !w || !out) return ADVOC_ERR_NULL;
  if (rows < 0 || k <= 0 || n <= 0) return ADVOC_ERR_BAD_SHAPE;
  if (rows == 0) return ADVOC_OK;
  const int64_t gx = advoc::ceil_div(rows, kTM);
  if (gx > 0x7fffffffLL) return ADVOC_ERR_UNSUPPORTED;
  dim3 grid((unsigned)gx, (unsigned)advoc::ceil_div(n, kTN));
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(matmul_nt_kernel, grid, dim3(256), 0, advoc::as_stream(stream), x, w, out,
                     rows, k, n);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

extern "C" int advoc_mel_dbnorm_f32(float* v, int64_t count, float min_level, float ref_db,
                                    float min_db, advoc_stream_t stream) {
  if (!v) return ADVOC_ERR_NULL;
  if (count < 0 || min_db >= 0.f) return ADVOC_ERR_BAD_SHAPE;
  if (count == 0) return ADVOC_OK;
  const int64_t blocks = advoc::ceil_div(count, 256);
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(mel_dbnorm_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256),
                     0, advoc::as_stream(stream), v, count, min_level, ref_db, min_db);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}
