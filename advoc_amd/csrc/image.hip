// Operand images of the split-bf16 matrix path: an fp32 tensor written ONCE as three bf16 planes
// (x = x0 + x1 + x2 exactly, split3 in x6.h), so that the GEMM kernels (igemm_x6d.hip) stream bf16 operands
// straight into LDS (buffer_load ... lds) instead of splitting every element again for every tap and every
// column tile that reads it.
//
//   activation image  [pixel (n, h, w_pitch)][c / 16][plane][16]   of  act(scale * x + shift) * mask * mask_scale
//                     -- exactly the value the fp32 loaders of igemm.hip feed the matrix cores, i.e. the fused
//                     input transform of a layer of models/advoc/advoc_model.py (lrelu :86-87,109; relu :138,155;
//                     the batch-norm affine :77-84; dropout behind it :144-149)
//   weight image      [tap][n][k / 16][plane][16]                  contraction axis innermost, from either kernel layout
//                     ([kh,kw,ci,co] of tf.layers.conv2d, [kh,kw,co,ci] of conv2d_transpose)
// The unit of both is the 96-byte K SLICE: the three planes of 16 consecutive contraction slots side by side, which is
// what one row of a GEMM K tile consumes.  (A plane-major layout was built first: one K tile then touched three
// 32-byte pieces in three distant cache lines per row, 3x the L1 / TA line traffic and a 4x longer L2 reuse distance
// -- 25-28 % L2 misses against 2 % for the fp32 loader, and slower than it.)
//
// Both kernels are HBM-bound elementwise passes: 4 B read + 6 B written per element.
#include "common.h"
#include "x6.h"

namespace advoc {
namespace {

// one thread = 8 consecutive channels of one pixel (c % 8 == 0): two float4 loads, three 16-byte stores
__global__ __launch_bounds__(256) void split_image_kernel(const float* __restrict__ x, uint16_t* __restrict__ img,
                                                          int64_t n8, int c, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float slope,
                                                          const uint8_t* __restrict__ mask, float mask_scale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const int64_t e = i * 8;
    float v[8];
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
    unsigned h0[8], h1[8], h2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split3(v[j], h0[j], h1[j], h2[j]);
    // element e = 16 s + 8 half + j  ->  slice s (48 uint16), plane pl at 16 pl, half at 8 half
    uint16_t* o = img + (e >> 4) * 48 + ((e >> 3) & 1) * 8;
    *reinterpret_cast<uint4*>(o) = make_uint4(pack_hi16(h0[0], h0[1]), pack_hi16(h0[2], h0[3]),
                                              pack_hi16(h0[4], h0[5]), pack_hi16(h0[6], h0[7]));
    *reinterpret_cast<uint4*>(o + 16) = make_uint4(pack_hi16(h1[0], h1[1]), pack_hi16(h1[2], h1[3]),
                                                   pack_hi16(h1[4], h1[5]), pack_hi16(h1[6], h1[7]));
    *reinterpret_cast<uint4*>(o + 32) = make_uint4(pack_hi16(h2[0], h2[1]), pack_hi16(h2[2], h2[3]),
                                                   pack_hi16(h2[4], h2[5]), pack_hi16(h2[6], h2[7]));
  }
}

// One workgroup = one 32 (k) x 32 (n) tile of one tap, through LDS so that both the fp32 reads (along n
// for the [tap][k][n] layout, along k for [tap][n][k]) and the bf16 writes (along k) are contiguous.
// sliced != 0: [tap][n][k / 16][plane][16] (igemm_x6d.hip); else [plane][tap][n][k] (register-split path of igemm.hip)
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ w, uint16_t* __restrict__ wq,
                                                            int taps, int n_total, int n_valid, int ktot, int b_kn,
                                                            int sliced) {
  __shared__ float tile[32][33];
  const int tk = (ktot + 31) / 32, tn = (n_total + 31) / 32;
  const int64_t plane = (int64_t)taps * n_total * ktot;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
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
        unsigned h0, h1, h2;
        split3(tile[tx][ty + 8 * i], h0, h1, h2);
        const int64_t o = ((int64_t)t * n_total + n) * ktot + k;
        if (sliced) {
          uint16_t* q = wq + (o >> 4) * 48 + (o & 15);     // ktot % 16 == 0: slices never straddle rows
          q[0] = (uint16_t)(h0 >> 16);
          q[16] = (uint16_t)(h1 >> 16);
          q[32] = (uint16_t)(h2 >> 16);
        } else {
          wq[o] = (uint16_t)(h0 >> 16);
          wq[plane + o] = (uint16_t)(h1 >> 16);
          wq[2 * plane + o] = (uint16_t)(h2 >> 16);
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace

int launch_split_image(const float* x, uint16_t* img, int64_t elems, int c, const float* scale, const float* shift,
                       int act, const uint8_t* mask, float mask_scale, hipStream_t stream) {
  if (!x || !img) return ADVOC_ERR_NULL;
  if (elems <= 0) return ADVOC_OK;
  if (c % 16 || elems % 16) return ADVOC_ERR_UNSUPPORTED;
  const int64_t n8 = elems / 8;
  int64_t blocks = ceil_div(n8, 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  const float slope = act == ADVOC_ACT_LRELU02 ? 0.2f : (act == ADVOC_ACT_RELU ? 0.f : 1.f);
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(split_image_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, img, n8, c, scale, shift,
                     slope, mask, mask_scale);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

int launch_split_weights(const float* w, uint16_t* wq, int taps, int n_total, int n_valid, int ktot, bool b_kn,
                         bool sliced, hipStream_t stream) {
  if (sliced && ktot % 16) return ADVOC_ERR_UNSUPPORTED;
  int64_t blocks = (int64_t)taps * ((ktot + 31) / 32) * ((n_total + 31) / 32);
  if (blocks > 4096) blocks = 4096;
  ADVOC_CLEAR_LAUNCH_ERROR();
  hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w, wq, taps, n_total,
                     n_valid, ktot, b_kn ? 1 : 0, sliced ? 1 : 0);
  ADVOC_RETURN_IF_LAUNCH_FAILED();
  return ADVOC_OK;
}

}  // namespace advoc
