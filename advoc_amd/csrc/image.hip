// Operand images of the split matrix paths: an fp32 tensor rewritten ONCE in the form the matrix cores consume,
// so that the GEMM kernels stream operands straight into LDS instead of transforming every element again for
// every tap and every column tile that reads it.
//
// (1) fp16 pair images (igemm_h3.hip).  x * 2^s = h0 + h1 + e with h0, h1 fp16 (round to nearest) and
//     |e| <= 2^-22 |x 2^s|; s is ONE power of two per GEMM operand, chosen from the operand's largest magnitude so
//     that it lands in [2^13, 2^14) (fp16 overflows at 65504; elements more than 2^17 below the largest lose
//     relative precision only: their absolute error stays <= 2^-39 of the largest).  An fp32 product a b is then
//     accumulated as THREE fp16 MFMA products a0 b1 + a1 b0 + a0 b0 with fp32 accumulation (fp16 x fp16 products
//     are exact in fp32; the dropped a1 b1 is <= 2^-22 of the product) and the accumulator is multiplied by the
//     exact 2^-(sa + sb) at the end: fp32-level error (measured at or below the fp32 MFMA chain against float64,
//     tools/micro/h3_numerics.py, tests/test_hip_conv.py) at 3 / 16 of the fp32 MFMA cost.
//       activation image  [pixel (n, h, w_pitch)][c / 32][plane][32]  of  act(scale * x + shift) * mask * mask_scale
//                         -- exactly the value the fp32 loaders of igemm.hip feed the matrix cores, i.e. the
//                         fused input transform of a layer of models/advoc/advoc_model.py (lrelu :86-87,109; relu
//                         :138,155; the batch-norm affine :77-84; dropout behind it :144-149)
//       weight image      [tap][n][k / 32][plane][32]                 contraction axis innermost, from either
//                         kernel layout ([kh,kw,ci,co] of tf.layers.conv2d, [kh,kw,co,ci] of conv2d_transpose)
//     The unit of both is the 128-byte K SLICE: the two planes of 32 consecutive contraction slots side by side
//     = ONE cache line = what one row of a GEMM K tile consumes.  That is the point of the layout: the L2 serves
//     a fixed number of line requests per clock, and a loader that uses 32-96 bytes of every 128-byte line it asks
//     for (fp32 rows of 16 channels, separate bf16 planes, 96-byte slices -- all three were built and measured)
//     saturates the L2 request rate at a third to a half of the useful bandwidth.
// (2) bf16 triple weights for the register-split kernels of igemm.hip (x6.h), plane-major.
//
// All kernels here are HBM-bound elementwise passes.
#include <hip/hip_fp16.h>

#include "common.h"
#include "x6.h"

namespace advoc {
namespace {

__host__ __device__ __forceinline__ float slope_of(int act) {
  return act == ADVOC_ACT_LRELU02 ? 0.2f : (act == ADVOC_ACT_RELU ? 0.f : 1.f);
}

// the 8 transformed values starting at element e (channels innermost, c % 8 == 0)
__device__ __forceinline__ void load8(const float* __restrict__ x, int64_t e, int c, const float* __restrict__ scale,
                                      const float* __restrict__ shift, float slope, const uint8_t* __restrict__ mask,
                                      float mask_scale, float v[8]) {
  const float4 a = *reinterpret_cast<const float4*>(x + e);
  const float4 b = *reinterpret_cast<const float4*>(x + e + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  if (scale) {
    const int ch = (int)(e % c);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], scale[ch + j], shift[ch + j]);
  }
  if (slope != 1.f) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], slope * v[j]);
  }
  if (mask) {
    const uint2 mk = *reinterpret_cast<const uint2*>(mask + e);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] *= (float)((mk.x >> (8 * j)) & 0xffu) * mask_scale;
      v[4 + j] *= (float)((mk.y >> (8 * j)) & 0xffu) * mask_scale;
    }
  }
}

// Largest-magnitude words are raised by ONE thread per workgroup, and only when the value it reads from memory is
// smaller: same-address atomics retire one by one in the L2 (measured ~10 ns each, tools/micro/img_bw.hip: 4096
// workgroups ending together = a 40 us tail on a 25 us pass).  A stale read can only be smaller than the truth (the word
// grows monotonically), so a skipped atomic is never a lost maximum.  The launchers keep these grids at kReduceBlocks.
__device__ __forceinline__ void raise_amax(unsigned* word, float m) {
  const unsigned bits = __float_as_uint(m);
  if (m > 0.f && bits > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(word, bits);
}

// amax[0] = max(amax[0], max |transformed x|) as the bit pattern of a non-negative float (orders like an unsigned)
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, int64_t n8, int c,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   float slope, const uint8_t* __restrict__ mask, float mask_scale,
                                                   unsigned* __restrict__ amax) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    float v[8];
    load8(x, i * 8, c, scale, shift, slope, mask, mask_scale, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[j]));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    raise_amax(amax, m);
  }
}

// weights: plain max |w| (any layout)
__global__ __launch_bounds__(256) void amax_flat_kernel(const float* __restrict__ w, int64_t n, unsigned* __restrict__ amax) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float m = 0.f;
  // 16-byte loads over the aligned body (weight tensors come from 16-byte aligned arena entries; checked), the tail
  // element by element
  const int64_t n4 = (reinterpret_cast<uintptr_t>(w) & 15) == 0 ? n >> 2 : 0;
  const float4* w4 = reinterpret_cast<const float4*>(w);
  for (int64_t i = tid; i < n4; i += stride) {
    const float4 v = w4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  for (int64_t i = 4 * n4 + tid; i < n; i += stride) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    raise_amax(amax, m);
  }
}

// out[seg] = max |base[off[seg] .. off[seg] + size[seg])| as float bits, for many tensors of one arena in ONE launch
// (blockIdx.y = tensor, blockIdx.x = slice of it); out zeroed by the launcher
__global__ __launch_bounds__(256) void segmented_amax_kernel(const float* __restrict__ base, const int64_t* __restrict__ off,
                                                             const int64_t* __restrict__ size, unsigned* __restrict__ out) {
  const int seg = blockIdx.y;
  const float* w = base + off[seg];
  const int64_t n = size[seg];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float m = 0.f;
  const int64_t n4 = (reinterpret_cast<uintptr_t>(w) & 15) == 0 ? n >> 2 : 0;
  const float4* w4 = reinterpret_cast<const float4*>(w);
  for (int64_t i = tid; i < n4; i += stride) {
    const float4 v = w4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  for (int64_t i = 4 * n4 + tid; i < n; i += stride) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    raise_amax(out + seg, m);
  }
}

// power of two that takes the largest magnitude into [2^13, 2^14); 1 for an all-zero (or non-finite) operand
__device__ __forceinline__ float up_scale(unsigned amax_bits) {
  const int e = (int)((amax_bits >> 23) & 0xffu);           // biased exponent of the largest magnitude
  if (e == 0 || e == 255) return 1.f;
  int s = 13 - (e - 127);
  s = s > 120 ? 120 : (s < -120 ? -120 : s);
  return __uint_as_float((unsigned)(s + 127) << 23);
}

// Delayed scaling (DELAYED): the scale comes from the largest magnitude of the PREVIOUS image written to this buffer
// (hdr[2]), placed at [2^9, 2^10) -- 2^6 of head room before fp16 overflows, and still 22 significant bits for every
// element within 2^12 of the largest -- while this pass records its own largest magnitude in hdr[0] for the next one:
// one pass over the tensor instead of two.  Values beyond the head room are counted in hdr[3] (and written saturated at
// +-65504): refit_image_kernel, always launched behind the pass, then rebuilds the image with the exact scale.
__device__ __forceinline__ float up_scale_delayed(unsigned prev_bits) {
  const int e = (int)((prev_bits >> 23) & 0xffu);
  if (e == 0 || e == 255) return 1.f;
  int s = 9 - (e - 127);
  s = s > 120 ? 120 : (s < -120 ? -120 : s);
  return __uint_as_float((unsigned)(s + 127) << 23);
}

// hdr[0] <- 0, hdr[2] <- old hdr[0]: the image about to be written becomes "current", the last one "previous";
// hdr[4] <- hdr[3]: the out-of-range count before this pass (refit_image_kernel compares the two)
__global__ void rotate_hdr_kernel(unsigned* __restrict__ hdr) {
  hdr[2] = hdr[0];
  hdr[0] = 0u;
  hdr[4] = hdr[3];
}

// Does the one-pass (delayed-scale) image just written under this header have to be rebuilt with its own scale?
//   * a value left the fp16 range under the previous image's scale (hdr[3] moved during the pass), or
//   * the tensor shrank by more than 2^6 (the small elements would keep absolute, not relative, precision), or
//   * there was no usable previous magnitude (all-zero / non-finite previous image) and this one is not all zero.
// Inside [2^-6, 2^6] of the previous magnitude the one-pass image keeps >= 22 significant bits for every element within
// 2^-9 of the largest and an absolute error <= 2^-28 of the largest below that.
__device__ __forceinline__ bool image_needs_refit(const unsigned* __restrict__ hdr) {
  const unsigned cur = __hip_atomic_load(hdr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned prev = __hip_atomic_load(hdr + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int ec = (int)((cur >> 23) & 0xffu), ep = (int)((prev >> 23) & 0xffu);
  if (__hip_atomic_load(hdr + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) !=
      __hip_atomic_load(hdr + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;
  if (ep == 0 || ep == 255) return ec != 0;
  return ec + 6 < ep;
}

// out[ch] += sum over the replicas of table[r][ch]
__global__ void colsum_reduce_kernel(const float* __restrict__ table, float* __restrict__ out, int c) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  float t = 0.f;
  for (int r = 0; r < kColsumReplicas; ++r) t += table[(size_t)r * c + ch];
  out[ch] += t;
}

// one thread = 8 consecutive channels of one pixel: two float4 loads, two 16-byte stores.
// hdr[0] = amax bits (input; DELAYED: accumulated here), hdr[1] = 2^-s as float bits (output, written by the first thread).
template <bool DELAYED, bool COLSUM = false>
__global__ __launch_bounds__(256) void pair_image_kernel(const float* __restrict__ x, __half* __restrict__ img,
                                                         int64_t n8, int c, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float slope,
                                                         const uint8_t* __restrict__ mask, float mask_scale,
                                                         unsigned* __restrict__ hdr, float* __restrict__ colsum,
                                                         int w_log, int w_pitch) {
  const float up = DELAYED ? up_scale_delayed(hdr[2]) : up_scale(hdr[0]);
  // colsum != null: also accumulate the per-channel sums of the transformed values over the LOGICAL pixels (the bias
  // gradient of the layer whose output gradient this is, models/advoc/advoc_model.py: tf.gradients w.r.t. the conv
  // biases) -- the launcher guarantees that the block size and the grid stride are multiples of c / 8, so a thread
  // keeps ONE group of 8 channels for the whole loop
  // (COLSUM is a template parameter: the plain instances keep the registers and the code of the image pass alone)
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // logical column of this thread's pixel, advanced without divisions: the pixel index grows by stride / (c / 8) per trip
  int xcol = 0, xstep = 0;
  if (COLSUM) {
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int groups = c >> 3;
    xcol = (int)((i0 / groups) % w_pitch);
    xstep = (int)(((int64_t)gridDim.x * blockDim.x / groups) % w_pitch);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) hdr[1] = __float_as_uint(1.f / up);     // exact: a power of two
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float vmax = 0.f;
  int sat = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const int64_t e = i * 8;
    float v[8];
    load8(x, e, c, scale, shift, slope, mask, mask_scale, v);
    if (COLSUM) {
      if (xcol < w_log) {
#pragma unroll
        for (int j = 0; j < 8; ++j) cs[j] += v[j];
      }
      xcol += xstep;
      if (xcol >= w_pitch) xcol -= w_pitch;
    }
    if (DELAYED) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        vmax = fmaxf(vmax, fabsf(v[j]));
        const float a = v[j] * up;
        if (fabsf(a) > 65504.f) { v[j] = copysignf(65504.f, a) / up; ++sat; }
      }
    }
    __half2 h0[4], h1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = v[2 * j] * up, b = v[2 * j + 1] * up;
      const __half a0 = __float2half_rn(a), b0 = __float2half_rn(b);
      h0[j] = __halves2half2(a0, b0);
      h1[j] = __halves2half2(__float2half_rn(a - __half2float(a0)), __float2half_rn(b - __half2float(b0)));
    }
    // element e = 32 s + 8 q + j  ->  slice s (64 halves), plane pl at 32 pl, octet q at 8 q
    __half* o = img + (e >> 5) * 64 + ((e >> 3) & 3) * 8;
    *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(h0);
    *reinterpret_cast<uint4*>(o + 32) = *reinterpret_cast<const uint4*>(h1);
  }
  if (DELAYED) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      vmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
      raise_amax(hdr, vmax);
    }
    if (sat) atomicAdd(hdr + 3, (unsigned)sat);
  }
  if (COLSUM) {
    __shared__ float s_cs[256][9];
#pragma unroll
    for (int j = 0; j < 8; ++j) s_cs[threadIdx.x][j] = cs[j];
    __syncthreads();
    const int groups = c >> 3;                       // <= 128, divides 256
    if ((int)threadIdx.x < groups) {
      float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int m = threadIdx.x; m < 256; m += groups)
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] += s_cs[m][j];
#pragma unroll
      // (into one of kColsumReplicas copies: thousands of blocks adding to the SAME c addresses serialise in the L2 --
      // measured 0.2 ms per launch; colsum_reduce_kernel folds the copies afterwards)
      for (int j = 0; j < 8; ++j)
        unsafeAtomicAdd(colsum + (size_t)(blockIdx.x & (kColsumReplicas - 1)) * c + threadIdx.x * 8 + j, t[j]);
    }
  }
}

// Always launched behind a delayed-scale image pass; exits at once unless image_needs_refit().  Then it rebuilds the
// image (both sources of the operand: blockIdx.y) with the EXACT scale of the magnitude the pass has just recorded in
// hdr[0] -- the two-pass form, its first pass already done -- so that no consumer ever reads a clamped or underflowed
// operand: the step that sees a tensor jump is as exact as any other, without a host round trip.  hdr[5] counts the
// refits (summaries).  The bias-gradient sums of the pass are taken from the fp32 values and need no repair.
struct RefitSource {
  const float* x;
  __half* img;
  int64_t n8;
  int c;
  const float* scale;
  const float* shift;
  float slope;
  const uint8_t* mask;
  float mask_scale;
};
// The LAST workgroup to finish also rotates the header for the next pass (hdr[2] <- hdr[0], hdr[0] <- 0, hdr[4] <- hdr[3];
// arrival counter in hdr[6]): every workgroup has read what it needs by then, and the one-thread rotate_hdr_kernel
// launch in front of every one-pass image (44 per train step, ~5 us each) is gone.
__device__ __forceinline__ void refit_done(unsigned* __restrict__ hdr) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned total = gridDim.x * gridDim.y;
    if (atomicAdd(hdr + 6, 1u) == total - 1) {
      hdr[6] = 0u;
      hdr[2] = hdr[0];
      hdr[4] = hdr[3];
      __hip_atomic_store(hdr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
__global__ __launch_bounds__(256) void refit_image_kernel(RefitSource s0, RefitSource s1, unsigned* __restrict__ hdr) {
  // (volatile reads: the decision must come from memory before this workgroup is counted as done)
  const bool needs = image_needs_refit(hdr);
  const unsigned cur_bits = __hip_atomic_load(hdr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!needs) { refit_done(hdr); return; }
  const RefitSource& s = blockIdx.y == 0 ? s0 : s1;
  const float up = up_scale(cur_bits);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    hdr[1] = __float_as_uint(1.f / up);
    atomicAdd(hdr + 5, 1u);
  }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < s.n8; i += stride) {
    const int64_t e = i * 8;
    float v[8];
    load8(s.x, e, s.c, s.scale, s.shift, s.slope, s.mask, s.mask_scale, v);
    __half2 h0[4], h1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = v[2 * j] * up, b = v[2 * j + 1] * up;
      const __half a0 = __float2half_rn(a), b0 = __float2half_rn(b);
      h0[j] = __halves2half2(a0, b0);
      h1[j] = __halves2half2(__float2half_rn(a - __half2float(a0)), __float2half_rn(b - __half2float(b0)));
    }
    __half* o = s.img + (e >> 5) * 64 + ((e >> 3) & 3) * 8;
    *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(h0);
    *reinterpret_cast<uint4*>(o + 32) = *reinterpret_cast<const uint4*>(h1);
  }
  refit_done(hdr);
}

// One workgroup = one 32 (k) x 32 (n) tile of one tap, through LDS so that both the fp32 reads (along n for the
// [tap][k][n] layout, along k for [tap][n][k]) and the 16-bit writes (along k) are contiguous.
//   pairs == 0: bf16 triple, wq[plane][tap][n_total][ktot]                 (register-split path of igemm.hip)
//   pairs != 0: fp16 pair,   wq[tap][n_total][ktot / 32][plane][32], scaled by the power of two from hdr[0]
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ w, uint16_t* __restrict__ wq,
                                                            int taps, int n_total, int n_valid, int ktot, int b_kn,
                                                            int pairs, unsigned* __restrict__ hdr,
                                                            const unsigned* __restrict__ amax_src) {
  __shared__ float tile[32][33];
  const int tk = (ktot + 31) / 32, tn = (n_total + 31) / 32;
  const int64_t plane = (int64_t)taps * n_total * ktot;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
  float up = 1.f;
  if (pairs) {
    up = up_scale(amax_src ? amax_src[0] : hdr[0]);       // amax_src: the largest |w| from advoc_segmented_amax_f32
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      hdr[1] = __float_as_uint(1.f / up);
      if (amax_src) hdr[0] = amax_src[0];     // (r5) word 0 of a weight header = max |w|: the a-priori bound of igemm_patch.hip reads it
    }
  }
  for (int b = blockIdx.x; b < taps * tk * tn; b += gridDim.x) {
    const int t = b / (tk * tn), r = b - t * (tk * tn);
    const int k0 = (r / tn) * 32, n0 = (r % tn) * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = ty + 8 * i;
      float x = 0.f;
      if (b_kn) {            // tile[k][n]: lanes along n
        const int k = k0 + row, n = n0 + tx;
        if (k < ktot && n < n_valid) x = w[((int64_t)t * ktot + k) * n_total + n];
        tile[row][tx] = x;
      } else {               // tile[k][n] filled from rows of n: lanes along k
        const int n = n0 + row, k = k0 + tx;
        if (k < ktot && n < n_valid) x = w[((int64_t)t * n_valid + n) * ktot + k];
        tile[tx][row] = x;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + ty + 8 * i, k = k0 + tx;
      if (n < n_total && k < ktot) {
        const float x = tile[tx][ty + 8 * i];
        const int64_t o = ((int64_t)t * n_total + n) * ktot + k;
        if (pairs) {
          const float a = x * up;
          const __half a0 = __float2half_rn(a);
          const __half a1 = __float2half_rn(a - __half2float(a0));
          uint16_t* q = wq + (o >> 5) * 64 + (o & 31);     // ktot % 32 == 0: slices never straddle rows
          q[0] = __half_as_ushort(a0);
          q[32] = __half_as_ushort(a1);
        } else {
          unsigned h0, h1, h2;
          split3(x, h0, h1, h2);
          wq[o] = (uint16_t)(h0 >> 16);
          wq[plane + o] = (uint16_t)(h1 >> 16);
          wq[2 * plane + o] = (uint16_t)(h2 >> 16);
        }
      }
    }
    __syncthreads();
  }
}

// Every weight image of a network in ONE launch (advoc_weight_images_f32): blockIdx.y = image, described by a row of 8
// int64 in device memory {w offset from base (elements), taps, n_total, ktot, b_kn, index into the magnitude table, byte
// offset of the image in the pool, index of its 4-word header}; same arithmetic and layout as split_weights_kernel's
// fp16-pair form, the scale from the arena's magnitude table (advoc_segmented_amax_f32).
// (r5) hdr_words >= 4 + kL1Taps: the kernel also leaves, in words 4 .. 4 + taps of the image's header, the per-tap maxima over
// the image's rows n of sum_k |w[tap][n][k]| (float bits, rounded up) -- the factors of the a-priori bounds of igemm_patch.hip --
// and taps / K in words 2 / 3.  A workgroup owns (tap, 32-row block) and walks its K tiles, so a row's sum is complete in
// registers when the walk ends: one atomicMax per workgroup item into a word the launcher has zeroed, no scratch, no second
// launch (a first version summed 2^40 fixed-point partials with 64-bit atomics per tile and folded them in a second kernel:
// +0.18 ms per train step, profiles/r05_a_steady_census.md).
constexpr int kL1Taps = 16;
__global__ __launch_bounds__(256) void weight_images_kernel(const float* __restrict__ base,
                                                            const unsigned* __restrict__ amax,
                                                            const int64_t* __restrict__ table,
                                                            char* __restrict__ pool, unsigned* __restrict__ hdrs,
                                                            int hdr_words) {
  __shared__ float tile[32][33];
  __shared__ float s_rowmax[8];
  const int64_t* row = table + 8 * (int64_t)blockIdx.y;
  const float* w = base + row[0];
  const int taps = (int)row[1], n_total = (int)row[2], ktot = (int)row[3], b_kn = (int)row[4];
  uint16_t* wq = reinterpret_cast<uint16_t*>(pool + row[6]);
  unsigned* hdr = hdrs + (int64_t)hdr_words * row[7];
  const bool want_l1 = hdr_words >= 4 + kL1Taps && taps <= kL1Taps;
  const int tk = ktot / 32, tn = (n_total + 31) / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float up = up_scale(amax[row[5]]);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    hdr[1] = __float_as_uint(1.f / up);
    hdr[0] = amax[row[5]];                    // (r5) max |w|, see split_weights_kernel
    if (hdr_words >= 4 + kL1Taps) { hdr[2] = want_l1 ? (unsigned)taps : 0u; hdr[3] = (unsigned)ktot; }
  }
  for (int item = blockIdx.x; item < taps * tn; item += gridDim.x) {
    const int t = item / tn, n0 = (item - t * tn) * 32;
    float sa[4] = {0.f, 0.f, 0.f, 0.f};       // this thread's share (k = k0 + tx) of sum_k |w| for rows n0 + ty + 8 i
    for (int kt = 0; kt < tk; ++kt) {
      const int k0 = kt * 32;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rr = ty + 8 * i;
        float x = 0.f;
        if (b_kn) {
          const int k = k0 + rr, n = n0 + tx;
          if (n < n_total) x = w[((int64_t)t * ktot + k) * n_total + n];
          tile[rr][tx] = x;
        } else {
          const int n = n0 + rr, k = k0 + tx;
          if (n < n_total) x = w[((int64_t)t * n_total + n) * ktot + k];
          tile[tx][rr] = x;
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = n0 + ty + 8 * i, k = k0 + tx;
        if (n < n_total) {
          const float x = tile[tx][ty + 8 * i];
          sa[i] += fabsf(x);
          const float a = x * up;
          const __half a0 = __float2half_rn(a);
          const __half a1 = __float2half_rn(a - __half2float(a0));
          const int64_t o = ((int64_t)t * n_total + n) * ktot + k;
          uint16_t* q = wq + (o >> 5) * 64 + (o & 31);
          q[0] = __half_as_ushort(a0);
          q[32] = __half_as_ushort(a1);
        }
      }
      __syncthreads();
    }
    if (want_l1) {
      // rows: fold the 32 k-lanes of a half wave, then the largest of the block's 32 rows, rounded UP (an upper bound is
      // what the consumers need: the additions above round to nearest, 1 + 2^-17 covers tk * 32 of them)
      float m = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = sa[i];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        m = fmaxf(m, v);
      }
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      if ((threadIdx.x & 63) == 0) s_rowmax[threadIdx.x >> 6] = m;
      __syncthreads();
      if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(s_rowmax[0], s_rowmax[1]), fmaxf(s_rowmax[2], s_rowmax[3])) * 1.00001f;
        atomicMax(hdr + 4 + t, __float_as_uint(m));
      }
      __syncthreads();
    }
  }
}

int grid_for(int64_t items, int per_block) {
  int64_t blocks = ceil_div(items, per_block);
  if (blocks > 256 * 16) blocks = 256 * 16;
  return (int)(blocks < 1 ? 1 : blocks);
}
// passes that end in one atomic per workgroup (raise_amax): two workgroups per CU stream at the same rate as sixteen
// (5.2 TB/s on 1 GB, tools/micro/img_bw.hip) and leave an eighth of the atomics
constexpr int kReduceBlocks = 512;
int grid_for_reduce(int64_t items, int per_block) {
  const int g = grid_for(items, per_block);
  return g > kReduceBlocks ? kReduceBlocks : g;
}

}  // namespace

int launch_amax(const float* x, int64_t elems, int c, const float* scale, const float* shift, int act,
                const uint8_t* mask, float mask_scale, unsigned* amax, hipStream_t stream) {
  if (!x || !amax) return ADVOC_ERR_NULL;
  if (elems <= 0) return ADVOC_OK;
  if (c % 8 || elems % 8) return ADVOC_ERR_UNSUPPORTED;
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(amax_kernel, dim3(grid_for_reduce(elems / 8, 256 * 4)), dim3(256), 0, stream, x, elems / 8, c, scale,
                     shift, slope_of(act), mask, mask_scale, amax);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

// max |x| over a flat fp32 array of any length, raised in *amax (float bits; the caller zeroes it)
int launch_amax_any(const float* x, int64_t elems, unsigned* amax, hipStream_t stream) {
  if (!x || !amax) return ADVOC_ERR_NULL;
  if (elems <= 0) return ADVOC_OK;
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(amax_flat_kernel, dim3(grid_for_reduce(elems, 256 * 16)), dim3(256), 0, stream, x, elems, amax);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

int launch_colsum_reduce(const float* table, float* out, int c, hipStream_t stream) {
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((c + 255) / 256), dim3(256), 0, stream, table, out, c);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

bool image_colsum_ok(int c) { return c >= 32 && c <= 1024 && 256 % (c / 8) == 0 && c % 8 == 0; }

int launch_pair_image(const float* x, uint16_t* img, int64_t elems, int c, const float* scale, const float* shift,
                      int act, const uint8_t* mask, float mask_scale, unsigned* hdr, bool delayed, hipStream_t stream,
                      float* colsum_out, int w_log, int w_pitch, float* colsum_table) {
  if (!x || !img || !hdr) return ADVOC_ERR_NULL;
  if (elems <= 0) return ADVOC_OK;
  if (c % 32 || elems % 32) return ADVOC_ERR_UNSUPPORTED;
  if (colsum_out && (!image_colsum_ok(c) || w_pitch <= 0 || !colsum_table)) return ADVOC_ERR_UNSUPPORTED;
  float* colsum = colsum_out ? colsum_table : nullptr;
  if (colsum) {
    hipError_t e = hipMemsetAsync(colsum_table, 0, sizeof(float) * kColsumReplicas * (size_t)c, stream);
    if (e != hipSuccess) { note_hip_error(e); return ADVOC_ERR_HIP; }
  }
  ADVOC_CLEAR_LAUNCH_ERROR();
  // (256 threads per block: block size and grid stride are multiples of c / 8 whenever image_colsum_ok(c))
  auto kern = delayed ? (colsum ? pair_image_kernel<true, true> : pair_image_kernel<true, false>)
                      : (colsum ? pair_image_kernel<false, true> : pair_image_kernel<false, false>);
  hipLaunchKernelGGL(kern, dim3(delayed ? grid_for_reduce(elems / 8, 256) : grid_for(elems / 8, 256)), dim3(256), 0, stream, x, reinterpret_cast<__half*>(img),
                     elems / 8, c, scale, shift, slope_of(act), mask, mask_scale, hdr, colsum, w_log, w_pitch);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  if (colsum) {
    ADVOC_CLEAR_LAUNCH_ERROR();
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((c + 255) / 256), dim3(256), 0, stream, colsum_table, colsum_out, c);
    ADVOC_RETURN_IF_LAUNCH_FAILED();
  }
  return ADVOC_OK;
}

// The always-launched check behind a one-pass image (built by pair_image_kernel<true> here, or by the producers'
// epilogues, image_emit.h): exact re-image when needed, header rotation by the last workgroup.
int launch_image_refit(const ImageSource& s0, const ImageSource& s1, uint16_t* img, unsigned* hdr, hipStream_t stream) {
  if (!img || !hdr || s0.elems <= 0) return ADVOC_OK;
  const int64_t b0 = (4 * s0.elems + 255) / 256 * 256;
  uint16_t* img1 = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(img) + b0);
  const RefitSource r0 = {s0.x, reinterpret_cast<__half*>(img), s0.elems / 8, s0.c, s0.scale, s0.shift, slope_of(s0.act),
                          s0.mask, s0.mask_scale};
  const RefitSource r1 = {s1.x, reinterpret_cast<__half*>(img1), s1.elems / 8, s1.c, s1.scale, s1.shift, slope_of(s1.act),
                          s1.mask, s1.mask_scale};
  ADVOC_CLEAR_LAUNCH_ERROR();
  // (64 workgroups per source: the common case is "nothing to do", and their arrival atomics are serial)
  hipLaunchKernelGGL(refit_image_kernel, dim3(64, s1.elems ? 2 : 1), dim3(256), 0, stream, r0, r1, hdr);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

// One GEMM operand (one or two channel-concatenated sources, ONE scale) -> image at `img` (source 1 behind source 0 at
// its 256-byte-rounded size) and header `hdr` (16 bytes, caller-owned, persistent across calls for delayed scaling).
int make_operand_image(const ImageSource& s0, const ImageSource& s1, uint16_t* img, unsigned* hdr, bool delayed,
                       hipStream_t stream, float* colsum0, int w_log, int w_pitch, float* colsum_table, bool keep_history) {
  const int64_t b0 = (4 * s0.elems + 255) / 256 * 256;
  uint16_t* img1 = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(img) + b0);
  int rc = ADVOC_OK;
  if (!delayed) {      // (delayed: the header was rotated at the end of the previous call on this buffer)
    hipError_t e = hipMemsetAsync(hdr, 0, 8, stream);
    if (e != hipSuccess) { note_hip_error(e); return ADVOC_ERR_HIP; }
    rc = launch_amax(s0.x, s0.elems, s0.c, s0.scale, s0.shift, s0.act, s0.mask, s0.mask_scale, hdr, stream);
    if (rc == ADVOC_OK && s1.elems)
      rc = launch_amax(s1.x, s1.elems, s1.c, s1.scale, s1.shift, s1.act, s1.mask, s1.mask_scale, hdr, stream);
    if (rc != ADVOC_OK) return rc;
  }
  rc = launch_pair_image(s0.x, img, s0.elems, s0.c, s0.scale, s0.shift, s0.act, s0.mask, s0.mask_scale, hdr, delayed, stream,
                         colsum0, w_log, w_pitch, colsum_table);
  if (rc == ADVOC_OK && s1.elems)
    rc = launch_pair_image(s1.x, img1, s1.elems, s1.c, s1.scale, s1.shift, s1.act, s1.mask, s1.mask_scale, hdr, delayed,
                           stream, nullptr, 0, 0, nullptr);
  if (rc == ADVOC_OK && delayed && s0.elems > 0) {
    rc = launch_image_refit(s0, s1, img, hdr, stream);
  } else if (rc == ADVOC_OK && s0.elems > 0 && keep_history) {
    // exact image under a persistent header: leave the header rotated, ready for a one-pass image next time (the per-call
    // images of the deep layers live in the launch workspace: r3 still launched this kernel behind each, 10 per step)
    ADVOC_CLEAR_LAUNCH_ERROR();
    hipLaunchKernelGGL(rotate_hdr_kernel, dim3(1), dim3(1), 0, stream, hdr);
    ADVOC_RETURN_IF_LAUNCH_FAILED();
  }
  return rc;
}

int launch_split_weights(const float* w, uint16_t* wq, int taps, int n_total, int n_valid, int ktot, bool b_kn,
                         hipStream_t stream) {
  int64_t blocks = (int64_t)taps * ((ktot + 31) / 32) * ((n_total + 31) / 32);
  if (blocks > 4096) blocks = 4096;
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w, wq, taps, n_total,
                     n_valid, ktot, b_kn ? 1 : 0, 0, (unsigned*)nullptr, (const unsigned*)nullptr);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

int launch_pair_weights(const float* w, uint16_t* wq, int taps, int n_total, int ktot, bool b_kn, unsigned* hdr,
                        hipStream_t stream, const unsigned* amax_src) {
  if (!w || !wq || !hdr) return ADVOC_ERR_NULL;
  if (ktot % 32) return ADVOC_ERR_UNSUPPORTED;
  const int64_t n = (int64_t)taps * n_total * ktot;
  if (!amax_src) {                      // (hdr[0] zeroed by the caller)
    ADVOC_CLEAR_LAUNCH_ERROR();
    hipLaunchKernelGGL(amax_flat_kernel, dim3(grid_for_reduce(n, 256 * 16)), dim3(256), 0, stream, w, n, hdr);
    ADVOC_RETURN_IF_LAUNCH_FAILED();
  }
  int64_t blocks = (int64_t)taps * (ktot / 32) * ((n_total + 31) / 32);
  if (blocks > 4096) blocks = 4096;
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w, wq, taps, n_total,
                     n_total, ktot, b_kn ? 1 : 0, 1, hdr, amax_src);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

}  // namespace advoc

extern "C" int advoc_weight_images_f32(const float* base, const uint32_t* amax, const int64_t* table, int32_t count,
                                       void* pool, uint32_t* hdrs, advoc_stream_t stream) {
  if (count < 0) return ADVOC_ERR_BAD_SHAPE;
  if (count == 0) return ADVOC_OK;
  if (!base || !amax || !table || !pool || !hdrs) return ADVOC_ERR_NULL;
  if (count > 65535) return ADVOC_ERR_UNSUPPORTED;
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(advoc::weight_images_kernel, dim3(256, (unsigned)count), dim3(256), 0, advoc::as_stream(stream), base,
                     amax, table, reinterpret_cast<char*>(pool), hdrs, 4);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

extern "C" int advoc_weight_images_l1_f32(const float* base, const uint32_t* amax, const int64_t* table, int32_t count,
                                          void* pool, uint32_t* hdrs, advoc_stream_t stream) {
  if (count < 0) return ADVOC_ERR_BAD_SHAPE;
  if (count == 0) return ADVOC_OK;
  if (!base || !amax || !table || !pool || !hdrs) return ADVOC_ERR_NULL;
  if (count > 65535) return ADVOC_ERR_UNSUPPORTED;
  // (the per-tap maxima are raised with atomicMax: every header starts from zero)
  hipError_t e = hipMemsetAsync(hdrs, 0, sizeof(uint32_t) * ADVOC_WEIGHT_HDR_L1_WORDS * (size_t)count, advoc::as_stream(stream));
  if (e != hipSuccess) { advoc::note_hip_error(e); return ADVOC_ERR_HIP; }
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(advoc::weight_images_kernel, dim3(256, (unsigned)count), dim3(256), 0, advoc::as_stream(stream), base,
                     amax, table, reinterpret_cast<char*>(pool), hdrs, ADVOC_WEIGHT_HDR_L1_WORDS);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

extern "C" int advoc_segmented_amax_f32(const float* base, const int64_t* offsets, const int64_t* sizes, int32_t count,
                                        uint32_t* amax_out, advoc_stream_t stream) {
  if (count < 0) return ADVOC_ERR_BAD_SHAPE;
  if (count == 0) return ADVOC_OK;
  if (!base || !offsets || !sizes || !amax_out) return ADVOC_ERR_NULL;
  if (count > 65535) return ADVOC_ERR_UNSUPPORTED;
  hipStream_t s = advoc::as_stream(stream);
  hipError_t e = hipMemsetAsync(amax_out, 0, sizeof(uint32_t) * (size_t)count, s);
  if (e != hipSuccess) { advoc::note_hip_error(e); return ADVOC_ERR_HIP; }
  ADVOC_CLEAR_LAUNCH_ERROR();
  // 256 slices per tensor: the 8M-element kernels of the deep layers set the duration of the launch (64 slices: 100 us)
  hipLaunchKernelGGL(advoc::segmented_amax_kernel, dim3(256, (unsigned)count), dim3(256), 0, s, base, offsets, sizes,
                     amax_out);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}
