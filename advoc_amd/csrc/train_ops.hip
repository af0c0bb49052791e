// Loss, optimiser and dropout-mask kernels of the AdVoc train step (gfx950).  All HBM-bound
// streaming kernels: float4 loads, wave-shuffle reductions, one atomic per block.
//
// Reference ops replaced
//   advoc_gan_d_loss / advoc_gan_g_loss : Sigmoid, Log, Abs, Mean and their gradients,
//                                         models/advoc/advoc_model.py:201,238-245
//   advoc_adam_tf_f32                   : tf.train.AdamOptimizer(0.0002, 0.5) ApplyAdam, :250-257
//   advoc_dropout_mask_u8               : RandomUniform + Floor of tf.nn.dropout, :144-149
#include "common.h"

namespace {

using advoc::wave_sum;

constexpr float kEps = 1e-12f;  // EPS, advoc_model.py:8

__device__ __forceinline__ float sigmoidf(float z) { return 1.f / (1.f + expf(-z)); }

// the discriminator's output activation on its own (advoc_model.py:201; the train step fuses it into the loss kernels)
__global__ __launch_bounds__(256) void sigmoid_kernel(const float* __restrict__ z, float* __restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = sigmoidf(z[i]);
}

// block-level sum of `v` added atomically into *dst (one atomic per block)
__device__ __forceinline__ void block_atomic_sum(float v, float* dst, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    unsafeAtomicAdd(dst, s);
  }
  __syncthreads();
}

// discrim_loss = mean(-(log(p_real + EPS) + log(1 - p_fake + EPS)))
__global__ __launch_bounds__(256) void d_loss_kernel(const float* __restrict__ zr,
                                                     const float* __restrict__ zf, int64_t n,
                                                     float inv_n, float* __restrict__ dzr,
                                                     float* __restrict__ dzf, float* __restrict__ sums) {
  __shared__ float red[4];
  float s = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float pr = sigmoidf(zr[i]), pf = sigmoidf(zf[i]);
    s += -(logf(pr + kEps) + logf(1.f - pf + kEps));
    // d/dz of -log(sigmoid(z) + EPS) = -p(1-p)/(p+EPS);  of -log(1 - sigmoid(z) + EPS) = p(1-p)/(1-p+EPS)
    if (dzr) dzr[i] = -inv_n * pr * (1.f - pr) / (pr + kEps);
    if (dzf) dzf[i] = inv_n * pf * (1.f - pf) / (1.f - pf + kEps);
  }
  block_atomic_sum(s, sums, red);
}

// gen_loss = gan_w * mean(-log(p_fake + EPS)) + l1_w * mean(|target - gen|)
__global__ __launch_bounds__(256) void g_gan_kernel(const float* __restrict__ zf, int64_t n, float scale,
                                                    float* __restrict__ dzf, float* __restrict__ sums) {
  __shared__ float red[4];
  float s = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float pf = sigmoidf(zf[i]);
    s += -logf(pf + kEps);
    if (dzf) dzf[i] = -scale * pf * (1.f - pf) / (pf + kEps);
  }
  block_atomic_sum(s, sums, red);
}

__global__ __launch_bounds__(256) void l1_kernel(const float* __restrict__ gen,
                                                 const float* __restrict__ target, int64_t n,
                                                 float scale, int accum, float* __restrict__ dgen,
                                                 float* __restrict__ sums) {
  __shared__ float red[4];
  float s = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float d = target[i] - gen[i];
    s += fabsf(d);
    if (dgen) {
      // d|t - g|/dg = -sign(t - g), sign(0) = 0 (tf.abs gradient)
      const float g = scale * (d > 0.f ? -1.f : (d < 0.f ? 1.f : 0.f));
      dgen[i] = accum ? dgen[i] + g : g;
    }
  }
  block_atomic_sum(s, sums, red);
}

// TF ApplyAdam: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr_t * m / (sqrt(v) + eps),
// lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) computed by the caller (epsilon is NOT bias-corrected).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   int64_t n, float lr_t, float b1, float b2, float eps,
                                                   float gscale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      float4 pp = *reinterpret_cast<float4*>(p + i);
      const float4 gg = *reinterpret_cast<const float4*>(g + i);
      float4 mm = *reinterpret_cast<float4*>(m + i);
      float4 vv = *reinterpret_cast<float4*>(v + i);
#define ADVOC_ADAM1(c)                                   \
  {                                                      \
    const float gr = gg.c * gscale;                      \
    mm.c = b1 * mm.c + (1.f - b1) * gr;                  \
    vv.c = b2 * vv.c + (1.f - b2) * gr * gr;             \
    pp.c -= lr_t * mm.c / (sqrtf(vv.c) + eps);           \
  }
      ADVOC_ADAM1(x) ADVOC_ADAM1(y) ADVOC_ADAM1(z) ADVOC_ADAM1(w)
#undef ADVOC_ADAM1
      *reinterpret_cast<float4*>(p + i) = pp;
      *reinterpret_cast<float4*>(m + i) = mm;
      *reinterpret_cast<float4*>(v + i) = vv;
    } else {
      for (int64_t j = i; j < n; ++j) {
        const float gr = g[j] * gscale;
        m[j] = b1 * m[j] + (1.f - b1) * gr;
        v[j] = b2 * v[j] + (1.f - b2) * gr * gr;
        p[j] -= lr_t * m[j] / (sqrtf(v[j]) + eps);
      }
    }
  }
}

// Philox-4x32-10 counter RNG: element e of stream `seed` depends only on (seed, e), so masks are
// reproducible and independent of how a global batch is sharded over GPUs.
__device__ __forceinline__ uint4 philox4x32(uint4 ctr, uint2 key) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
    const uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
    ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
    key.x += W0;
    key.y += W1;
  }
  return ctr;
}

// mask[i] = floor(keep + u) with u uniform in [0,1): 1 when u >= 1 - keep  (tf.nn.dropout)
__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ mask, int64_t n,
                                                           uint64_t seed, uint64_t offset, float keep) {
  const int64_t quads = (n + 3) / 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const uint32_t thresh = (uint32_t)fminf(fmaxf((1.0f - keep) * 4294967296.0f, 0.f), 4294967295.f);
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += stride) {
    const uint64_t c = (offset >> 2) + (uint64_t)q;     // offset is a multiple of 4
    const uint4 r = philox4x32(make_uint4((uint32_t)c, (uint32_t)(c >> 32), 0u, 0u),
                               make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
    const int64_t i = q * 4;
    if (i + 4 <= n) {
      uchar4 o;
      o.x = rr[0] >= thresh; o.y = rr[1] >= thresh; o.z = rr[2] >= thresh; o.w = rr[3] >= thresh;
      *reinterpret_cast<uchar4*>(mask + i) = o;
    } else {
      for (int j = 0; i + j < n; ++j) mask[i + j] = rr[j] >= thresh;
    }
  }
}

unsigned grid_for(int64_t n, int per_thread = 1) {
  int64_t b = advoc::ceil_div(n, 256LL * per_thread);
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// the loss kernels end in one atomic per workgroup on ONE address: same-address atomics retire one by one (~10 ns each,
// tools/micro/img_bw.hip), so their grids stay at two workgroups per CU
unsigned grid_for_sum(int64_t n) {
  const unsigned b = grid_for(n);
  return b > 512 ? 512 : b;
}

}  // namespace

using advoc::as_stream;

extern "C" int advoc_gan_d_loss(const float* logit_real, const float* logit_fake, int64_t n,
                                float* dlogit_real, float* dlogit_fake, float* loss_sum,
                                advoc_stream_t stream) {
  if (!logit_real || !logit_fake || !loss_sum) return ADVOC_ERR_NULL;
  if (n <= 0) return ADVOC_ERR_BAD_SHAPE;
  hipError_t e = hipMemsetAsync(loss_sum, 0, sizeof(float), as_stream(stream));
  if (e != hipSuccess) { advoc::note_hip_error(e); return ADVOC_ERR_HIP; }
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(d_loss_kernel, dim3(grid_for_sum(n)), dim3(256), 0, as_stream(stream), logit_real,
                     logit_fake, n, 1.f / (float)n, dlogit_real, dlogit_fake, loss_sum);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

extern "C" int advoc_gan_g_loss(const float* logit_fake, int64_t n_logits, const float* gen,
                                const float* target, int64_t n_spec, float gan_weight,
                                float l1_weight, float* dlogit_fake, float* dgen, int32_t accum_dgen,
                                float* loss_sums, advoc_stream_t stream) {
  if (!gen || !target || !loss_sums) return ADVOC_ERR_NULL;
  if (n_spec <= 0 || (logit_fake && n_logits <= 0)) return ADVOC_ERR_BAD_SHAPE;
  hipError_t e = hipMemsetAsync(loss_sums, 0, 2 * sizeof(float), as_stream(stream));
  if (e != hipSuccess) { advoc::note_hip_error(e); return ADVOC_ERR_HIP; }
  if (logit_fake) {
    ADVOC_CLEAR_LAUNCH_ERROR();
    hipLaunchKernelGGL(g_gan_kernel, dim3(grid_for_sum(n_logits)), dim3(256), 0, as_stream(stream),
                       logit_fake, n_logits, gan_weight / (float)n_logits, dlogit_fake, loss_sums);
    ADVOC_RETURN_IF_LAUNCH_FAILED();
  }
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(l1_kernel, dim3(grid_for_sum(n_spec)), dim3(256), 0, as_stream(stream), gen, target,
                     n_spec, l1_weight / (float)n_spec, accum_dgen, dgen, loss_sums + 1);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

extern "C" int advoc_adam_tf_f32(float* param, const float* grad, float* m, float* v, int64_t count,
                                 float lr_t, float beta1, float beta2, float epsilon,
                                 float grad_scale, advoc_stream_t stream) {
  if (!param || !grad || !m || !v) return ADVOC_ERR_NULL;
  if (count < 0) return ADVOC_ERR_BAD_SHAPE;
  if (count == 0) return ADVOC_OK;
  if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) |
       reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15)
    return ADVOC_ERR_UNSUPPORTED;   // float4 path needs 16-byte aligned arenas
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(count, 4)), dim3(256), 0, as_stream(stream), param,
                     grad, m, v, count, lr_t, beta1, beta2, epsilon, grad_scale);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

namespace {
// 16-byte stores, two workgroups per CU: the zero fill of a gradient arena at the HBM write rate (the framework's generic
// fill kernel ran the generator's 217 MB arena at 1.4 TB/s: 157 us per step)
__global__ __launch_bounds__(256) void zero_kernel(float4* __restrict__ p, int64_t n4, float* __restrict__ tail, int ntail) {
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) p[i] = z;
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0.f;
}
}  // namespace

// The gradient arenas start every step from zero (tf.gradients sums into fresh tensors; here the weight- and bias-gradient
// kernels accumulate into the arena): one launch per arena.
extern "C" int advoc_zero_f32(float* dst, int64_t count, advoc_stream_t stream) {
  if (!dst) return ADVOC_ERR_NULL;
  if (count < 0) return ADVOC_ERR_BAD_SHAPE;
  if (count == 0) return ADVOC_OK;
  if (reinterpret_cast<uintptr_t>(dst) & 15) return ADVOC_ERR_UNSUPPORTED;
  const int64_t n4 = count / 4;
  int64_t blocks = advoc::ceil_div(n4 > 0 ? n4 : 1, 256);
  if (blocks > 512) blocks = 512;
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), reinterpret_cast<float4*>(dst), n4,
                     dst + 4 * n4, (int)(count - 4 * n4));
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

extern "C" int advoc_sigmoid_f32(const float* logits, float* prob, int64_t count, advoc_stream_t stream) {
  if (!logits || !prob) return ADVOC_ERR_NULL;
  if (count < 0) return ADVOC_ERR_BAD_SHAPE;
  if (count == 0) return ADVOC_OK;
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(sigmoid_kernel, dim3(grid_for(count, 1)), dim3(256), 0, as_stream(stream), logits, prob, count);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

extern "C" int advoc_dropout_mask_u8(uint8_t* mask, int64_t count, uint64_t seed, uint64_t offset,
                                     float keep_prob, advoc_stream_t stream) {
  if (!mask) return ADVOC_ERR_NULL;
  if (count < 0 || !(keep_prob > 0.f && keep_prob <= 1.f)) return ADVOC_ERR_BAD_SHAPE;
  if (offset & 3) return ADVOC_ERR_UNSUPPORTED;
  if (count == 0) return ADVOC_OK;
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(count, 4)), dim3(256), 0, as_stream(stream),
                     mask, count, seed, offset, keep_prob);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}
