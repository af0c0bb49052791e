// Batch normalisation in training mode (gfx950): tf.layers.batch_normalization(axis=3,
// epsilon=1e-5, training=True) of models/advoc/advoc_model.py:77-84,173-177 -- always batch
// statistics (biased variance over N*H*W), no moving averages on the path.
//
// The normalised tensor is never written: forward only produces the per-channel affine
// (scale = gamma * invstd, shift = beta - mean * scale) that the CONSUMING conv folds into its
// loads (advoc_conv_layer.in_scale / in_shift).  Backward turns dL/dy into dL/dz in place.
// HBM-bound streaming kernels: 16-byte loads, 4 loads in flight per lane, LDS + one atomic per
// block and channel.  Sums are taken about a per-channel pivot (the first pixel) so the
// variance does not cancel catastrophically in fp32.
#include "common.h"

namespace {

// s1[c] += sum_p (a[p,c] - pa[c]);  s2[c] += sum_p (a[p,c] - pa[c]) * (b[p,c] - pb[c])
__global__ __launch_bounds__(256) void colsum2_kernel(const float* __restrict__ a,
                                                      const float* __restrict__ b,
                                                      const float* __restrict__ pa,
                                                      const float* __restrict__ pb, int64_t npix,
                                                      int c, double* __restrict__ s1,
                                                      double* __restrict__ s2) {
  __shared__ float4 red1[256];
  __shared__ float4 red2[256];
  const int quads = c / 4;
  const int lanes_q = quads < 256 ? quads : 256;
  const int groups = 256 / lanes_q;
  const int tq = threadIdx.x % lanes_q, tg = threadIdx.x / lanes_q;
  for (int qbase = 0; qbase < quads; qbase += lanes_q) {
    const int q = qbase + tq;
    float4 u = make_float4(0.f, 0.f, 0.f, 0.f), v = u;
    if (tg < groups && q < quads) {
      float4 za = make_float4(0.f, 0.f, 0.f, 0.f), zb = za;
      if (pa) za = *reinterpret_cast<const float4*>(pa + 4 * q);
      if (pb) zb = *reinterpret_cast<const float4*>(pb + 4 * q);
      const int64_t step = (int64_t)gridDim.x * groups;
      for (int64_t pix = (int64_t)blockIdx.x * groups + tg; pix < npix; pix += 2 * step) {
        const int64_t o0 = pix * c + 4 * q;
        const bool two = pix + step < npix;
        const int64_t o1 = two ? o0 + step * c : o0;
        float4 x0 = *reinterpret_cast<const float4*>(a + o0);
        float4 y0 = *reinterpret_cast<const float4*>(b + o0);
        float4 x1 = *reinterpret_cast<const float4*>(a + o1);
        float4 y1 = *reinterpret_cast<const float4*>(b + o1);
        const float k = two ? 1.f : 0.f;
        x0.x -= za.x; x0.y -= za.y; x0.z -= za.z; x0.w -= za.w;
        y0.x -= zb.x; y0.y -= zb.y; y0.z -= zb.z; y0.w -= zb.w;
        x1.x = (x1.x - za.x) * k; x1.y = (x1.y - za.y) * k; x1.z = (x1.z - za.z) * k; x1.w = (x1.w - za.w) * k;
        y1.x -= zb.x; y1.y -= zb.y; y1.z -= zb.z; y1.w -= zb.w;
        u.x += x0.x + x1.x; u.y += x0.y + x1.y; u.z += x0.z + x1.z; u.w += x0.w + x1.w;
        v.x += x0.x * y0.x + x1.x * y1.x; v.y += x0.y * y0.y + x1.y * y1.y;
        v.z += x0.z * y0.z + x1.z * y1.z; v.w += x0.w * y0.w + x1.w * y1.w;
      }
    }
    red1[threadIdx.x] = u;
    red2[threadIdx.x] = v;
    __syncthreads();
    if (tg == 0 && q < quads) {
      // cross-thread and cross-block combination in double: the per-channel sums feed gradients
      // (d gamma = sum g * zhat) that cancel heavily; fp32 atomics made them order-dependent
      double t1[4] = {0, 0, 0, 0}, t2[4] = {0, 0, 0, 0};
      for (int k = 0; k < groups; ++k) {
        const float4 r1 = red1[k * lanes_q + tq], r2 = red2[k * lanes_q + tq];
        t1[0] += r1.x; t1[1] += r1.y; t1[2] += r1.z; t1[3] += r1.w;
        t2[0] += r2.x; t2[1] += r2.y; t2[2] += r2.z; t2[3] += r2.w;
      }
      for (int k = 0; k < 4; ++k) {
        unsafeAtomicAdd(s1 + 4 * q + k, t1[k]);
        unsafeAtomicAdd(s2 + 4 * q + k, t2[k]);
      }
    }
    __syncthreads();
  }
}

__global__ void bn_finalize_kernel(const float* __restrict__ pivot, const double* __restrict__ s1,
                                   const double* __restrict__ s2, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float inv_n, float eps, int c,
                                   float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean, float* __restrict__ invstd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  const double dd = s1[i] * (double)inv_n;       // E[z - pivot]
  const float var = (float)fmax(s2[i] * (double)inv_n - dd * dd, 0.0);
  const float m = (float)((double)pivot[i] + dd);
  const float is = 1.0f / sqrtf(var + eps);
  const float sc = gamma[i] * is;
  mean[i] = m;
  invstd[i] = is;
  scale[i] = sc;
  shift[i] = beta[i] - m * sc;
}

// g <- gamma*invstd * (g - s1/N - (z - mean) * invstd^2 * s2/N),  s1 = sum g, s2 = sum g (z - mean)
__global__ __launch_bounds__(256) void bn_backward_apply_kernel(
    const float* __restrict__ z, float* __restrict__ g, int64_t total4, int c,
    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
    const double* __restrict__ s1, const double* __restrict__ s2, float inv_n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int quads = c / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
    const int q = (int)(i % quads);
    const float4 zz = reinterpret_cast<const float4*>(z)[i];
    float4 gg = reinterpret_cast<float4*>(g)[i];
    const float4 ga = *reinterpret_cast<const float4*>(gamma + 4 * q);
    const float4 mu = *reinterpret_cast<const float4*>(mean + 4 * q);
    const float4 is = *reinterpret_cast<const float4*>(invstd + 4 * q);
    const float4 a1 = make_float4((float)s1[4 * q], (float)s1[4 * q + 1], (float)s1[4 * q + 2], (float)s1[4 * q + 3]);
    const float4 a2 = make_float4((float)s2[4 * q], (float)s2[4 * q + 1], (float)s2[4 * q + 2], (float)s2[4 * q + 3]);
    gg.x = ga.x * is.x * (gg.x - a1.x * inv_n - (zz.x - mu.x) * is.x * is.x * a2.x * inv_n);
    gg.y = ga.y * is.y * (gg.y - a1.y * inv_n - (zz.y - mu.y) * is.y * is.y * a2.y * inv_n);
    gg.z = ga.z * is.z * (gg.z - a1.z * inv_n - (zz.z - mu.z) * is.z * is.z * a2.z * inv_n);
    gg.w = ga.w * is.w * (gg.w - a1.w * inv_n - (zz.w - mu.w) * is.w * is.w * a2.w * inv_n);
    reinterpret_cast<float4*>(g)[i] = gg;
  }
}

__global__ void bn_param_grad_kernel(const double* __restrict__ s1, const double* __restrict__ s2,
                                     const float* __restrict__ invstd, int c, int accumulate,
                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  const float dg = (float)(s2[i] * (double)invstd[i]), db = (float)s1[i];
  dgamma[i] = accumulate ? dgamma[i] + dg : dg;
  dbeta[i] = accumulate ? dbeta[i] + db : db;
}

// pivot-shifted sums -> rank-independent (sum z, sum z^2): what cross-replica batch norm adds up
__global__ void bn_unpivot_kernel(const float* __restrict__ pivot, double* __restrict__ s1, double* __restrict__ s2,
                                  double n, int c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  const double p = (double)pivot[i], a = s1[i], b = s2[i];
  s1[i] = a + n * p;
  s2[i] = b + 2.0 * p * a + n * p * p;
}

__global__ void bn_finalize_sums_kernel(const double* __restrict__ s1, const double* __restrict__ s2,
                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                        double inv_n, float eps, int c, float* __restrict__ scale,
                                        float* __restrict__ shift, float* __restrict__ mean,
                                        float* __restrict__ invstd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  const double m = s1[i] * inv_n;
  const float var = (float)fmax(s2[i] * inv_n - m * m, 0.0);
  const float is = 1.0f / sqrtf(var + eps);
  const float sc = gamma[i] * is;
  mean[i] = (float)m;
  invstd[i] = is;
  scale[i] = sc;
  shift[i] = beta[i] - (float)m * sc;
}

int launch_colsum2(const float* a, const float* b, const float* pa, const float* pb, int64_t npix, int c,
                   double* s1, double* s2, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(s1, 0, sizeof(double) * (size_t)c, stream);
  if (e == hipSuccess) e = hipMemsetAsync(s2, 0, sizeof(double) * (size_t)c, stream);
  if (e != hipSuccess) { advoc::note_hip_error(e); return ADVOC_ERR_HIP; }
  const int quads = c / 4;
  const int groups = 256 / (quads < 256 ? quads : 256);
  int64_t blocks = advoc::ceil_div(npix, (int64_t)groups * 16);
  if (blocks > 256) blocks = 256;   // one per CU: the same-address atomics at the end cost ~40 ns per block
  if (blocks < 1) blocks = 1;
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(colsum2_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, b, pa, pb, npix, c,
                     s1, s2);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

bool bn_shape_ok(int64_t npix, int c) {
  // a thread owns 4 channels; up to 256 quads per pass (threads beyond groups * quads idle, so any
  // multiple of 4 channels works), more only in whole passes of 256
  const int quads = c / 4;
  return npix > 0 && c > 0 && c % 4 == 0 && (quads <= 256 || quads % 256 == 0);
}

}  // namespace

using advoc::as_stream;

extern "C" int advoc_bn_forward(const float* z, int64_t npix, int32_t c, const float* gamma,
                                const float* beta, float epsilon, float* scale, float* shift,
                                float* mean, float* invstd, float* work, advoc_stream_t stream) {
  if (!z || !gamma || !beta || !scale || !shift || !mean || !invstd || !work) return ADVOC_ERR_NULL;
  if (npix <= 0 || c <= 0) return ADVOC_ERR_BAD_SHAPE;
  if (!bn_shape_ok(npix, c)) return ADVOC_ERR_UNSUPPORTED;
  // pivot = first pixel (row 0 of z): just a pointer
  if (reinterpret_cast<uintptr_t>(work) & 7) return ADVOC_ERR_UNSUPPORTED;
  double* w1 = reinterpret_cast<double*>(work);
  int rc = launch_colsum2(z, z, z, z, npix, c, w1, w1 + c, as_stream(stream));
  if (rc != ADVOC_OK) return rc;
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((c + 255) / 256), dim3(256), 0, as_stream(stream), z, w1,
                     w1 + c, gamma, beta, 1.0f / (float)npix, epsilon, c, scale, shift, mean, invstd);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

extern "C" int advoc_bn_backward(const float* z, float* g, int64_t npix, int32_t c, const float* gamma,
                                 const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                 int32_t accumulate, float* work, advoc_stream_t stream) {
  if (!z || !g || !gamma || !mean || !invstd || !dgamma || !dbeta || !work) return ADVOC_ERR_NULL;
  if (npix <= 0 || c <= 0) return ADVOC_ERR_BAD_SHAPE;
  if (!bn_shape_ok(npix, c)) return ADVOC_ERR_UNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(work) & 7) return ADVOC_ERR_UNSUPPORTED;
  double* w1 = reinterpret_cast<double*>(work);
  int rc = launch_colsum2(g, z, nullptr, mean, npix, c, w1, w1 + c, as_stream(stream));
  if (rc != ADVOC_OK) return rc;
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(bn_param_grad_kernel, dim3((c + 255) / 256), dim3(256), 0, as_stream(stream), w1,
                     w1 + c, invstd, c, accumulate, dgamma, dbeta);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  const int64_t total4 = npix * (c / 4);
  int64_t blocks = advoc::ceil_div(total4, 256 * 4);
  if (blocks > 4096) blocks = 4096;
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(bn_backward_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), z,
                     g, total4, c, gamma, mean, invstd, w1, w1 + c, 1.0f / (float)npix);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

// ---------------------------------------------------------------------------------------------
// Split forms for cross-replica (synchronised) batch norm: the caller sums `work` (2c doubles) over
// the replicas between the two halves and passes the global pixel count to the second.
// advoc_bn_forward == forward_stats + forward_finalize(count = npix); likewise backward.
// ---------------------------------------------------------------------------------------------
extern "C" int advoc_bn_forward_stats(const float* z, int64_t npix, int32_t c, float* work, advoc_stream_t stream) {
  if (!z || !work) return ADVOC_ERR_NULL;
  if (npix <= 0 || c <= 0) return ADVOC_ERR_BAD_SHAPE;
  if (!bn_shape_ok(npix, c)) return ADVOC_ERR_UNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(work) & 7) return ADVOC_ERR_UNSUPPORTED;
  double* w1 = reinterpret_cast<double*>(work);
  int rc = launch_colsum2(z, z, z, z, npix, c, w1, w1 + c, as_stream(stream));
  if (rc != ADVOC_OK) return rc;
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(bn_unpivot_kernel, dim3((c + 255) / 256), dim3(256), 0, as_stream(stream), z, w1, w1 + c,
                     (double)npix, c);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

extern "C" int advoc_bn_forward_finalize(const float* work, int64_t count_total, int32_t c, const float* gamma,
                                         const float* beta, float epsilon, float* scale, float* shift,
                                         float* mean, float* invstd, advoc_stream_t stream) {
  if (!work || !gamma || !beta || !scale || !shift || !mean || !invstd) return ADVOC_ERR_NULL;
  if (count_total <= 0 || c <= 0) return ADVOC_ERR_BAD_SHAPE;
  if (reinterpret_cast<uintptr_t>(work) & 7) return ADVOC_ERR_UNSUPPORTED;
  const double* w1 = reinterpret_cast<const double*>(work);
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(bn_finalize_sums_kernel, dim3((c + 255) / 256), dim3(256), 0, as_stream(stream), w1, w1 + c,
                     gamma, beta, 1.0 / (double)count_total, epsilon, c, scale, shift, mean, invstd);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

extern "C" int advoc_bn_backward_stats(const float* z, const float* g, int64_t npix, int32_t c, const float* mean,
                                       const float* invstd, float* dgamma, float* dbeta, int32_t accumulate,
                                       float* work, advoc_stream_t stream) {
  if (!z || !g || !mean || !invstd || !dgamma || !dbeta || !work) return ADVOC_ERR_NULL;
  if (npix <= 0 || c <= 0) return ADVOC_ERR_BAD_SHAPE;
  if (!bn_shape_ok(npix, c)) return ADVOC_ERR_UNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(work) & 7) return ADVOC_ERR_UNSUPPORTED;
  double* w1 = reinterpret_cast<double*>(work);
  int rc = launch_colsum2(g, z, nullptr, mean, npix, c, w1, w1 + c, as_stream(stream));
  if (rc != ADVOC_OK) return rc;
  // the parameter gradients are THIS replica's contribution (they are summed with the other gradients)
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(bn_param_grad_kernel, dim3((c + 255) / 256), dim3(256), 0, as_stream(stream), w1, w1 + c,
                     invstd, c, accumulate, dgamma, dbeta);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

extern "C" int advoc_bn_backward_apply(const float* z, float* g, int64_t npix, int32_t c, const float* gamma,
                                       const float* mean, const float* invstd, const float* work,
                                       int64_t count_total, advoc_stream_t stream) {
  if (!z || !g || !gamma || !mean || !invstd || !work) return ADVOC_ERR_NULL;
  if (npix <= 0 || c <= 0 || count_total <= 0) return ADVOC_ERR_BAD_SHAPE;
  if (!bn_shape_ok(npix, c)) return ADVOC_ERR_UNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(work) & 7) return ADVOC_ERR_UNSUPPORTED;
  const double* w1 = reinterpret_cast<const double*>(work);
  const int64_t total4 = npix * (c / 4);
  int64_t blocks = advoc::ceil_div(total4, 256 * 4);
  if (blocks > 4096) blocks = 4096;
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(bn_backward_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), z, g,
                     total4, c, gamma, mean, invstd, w1, w1 + c, 1.0f / (float)count_total);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}
