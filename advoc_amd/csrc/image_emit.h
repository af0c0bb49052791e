// Operand images written by the PRODUCER of a tensor (r3): the forward epilogues of the image kernels (igemm_h3.hip,
// igemm_patch.hip) also write the fp16 pair image the CONSUMING layer's next contraction reads, with the consumer's
// activation applied and under the consumer operand's one-pass (delayed) scale -- exactly the bytes
// pair_image_kernel<true> (image.hip) would produce from the fp32 tensor, so the consumer's image pass (a read and a
// write of the whole tensor) disappears.  The fp32 tensor is still written: the backward pass gates on it, the thin
// kernels read it, and refit_image_kernel rebuilds the image from it when a value leaves the scale's window.
#pragma once
#include <hip/hip_fp16.h>

#include "common.h"
#include "igemm.h"

namespace advoc {

// power of two that puts the previous image's largest magnitude at [2^9, 2^10) (image.hip: up_scale_delayed)
__device__ __forceinline__ float emit_up_scale(unsigned prev_bits) {
  const int e = (int)((prev_bits >> 23) & 0xffu);
  if (e == 0 || e == 255) return 1.f;
  int s = 9 - (e - 127);
  s = s > 120 ? 120 : (s < -120 ? -120 : s);
  return __uint_as_float((unsigned)(s + 127) << 23);
}

// the 4 channels e .. e + 3 (e % 4 == 0) of one pixel, e = linear NHWC element index of the tensor
__device__ __forceinline__ void emit4(const ImgOut& o, float up, float4 v, unsigned e, float& vmax, int& sat) {
  float t[4] = {v.x, v.y, v.z, v.w};
  __half h0[4], h1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    t[j] = fmaxf(t[j], o.slope * t[j]);
    vmax = fmaxf(vmax, fabsf(t[j]));
    float a = t[j] * up;
    if (fabsf(a) > 65504.f) { a = copysignf(65504.f, a); ++sat; }      // (refit_image_kernel repairs the image)
    h0[j] = __float2half_rn(a);
    h1[j] = __float2half_rn(a - __half2float(h0[j]));
  }
  uint16_t* dst = o.img + (size_t)(e >> 5) * 64 + (e & 31);
  *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(h0);
  *reinterpret_cast<uint2*>(dst + 32) = *reinterpret_cast<const uint2*>(h1);
}

// The same through a buffer descriptor, for epilogues that must not branch around their stores (igemm_patch.hip: the
// number of stores between a load and its use has to be known at compile time): byte_off = byte offset of element e in the
// fp32 tensor, or an out-of-range value for rows without a pixel -- the descriptor's range check drops those stores (and
// every store when the launch has no such consumer: num_records 0) and `valid` keeps them out of the statistics.
__device__ __forceinline__ void emit4_buffer(__amdgpu_buffer_rsrc_t rs, float slope, float up, float4 v, unsigned byte_off,
                                             bool valid, float& vmax, int& sat) {
  float t[4] = {v.x, v.y, v.z, v.w};
  __half h0[4], h1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    t[j] = fmaxf(t[j], slope * t[j]);
    vmax = fmaxf(vmax, valid ? fabsf(t[j]) : 0.f);
    float a = t[j] * up;
    const bool over = fabsf(a) > 65504.f;
    a = over ? copysignf(65504.f, a) : a;
    sat += (over && valid) ? 1 : 0;
    h0[j] = __float2half_rn(a);
    h1[j] = __float2half_rn(a - __half2float(h0[j]));
  }
  // element e = byte_off / 4 -> halves (e >> 5) * 64 + (e & 31) -> bytes: ((e >> 5) * 64 + (e & 31)) * 2
  const unsigned e = byte_off >> 2;
  const unsigned ib = byte_off >= 0xffffff00u ? 0xffffff00u : (((e >> 5) << 6) + (e & 31)) * 2u;
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const uint2 p0 = *reinterpret_cast<const uint2*>(h0), p1 = *reinterpret_cast<const uint2*>(h1);
  __builtin_amdgcn_raw_buffer_store_b64((u32x2){p0.x, p0.y}, rs, ib, 0, 0);
  __builtin_amdgcn_raw_buffer_store_b64((u32x2){p1.x, p1.y}, rs, ib, 64, 0);
}

// once per wave, after its last emit4: the largest magnitude it saw and its out-of-window count into the header
__device__ __forceinline__ void emit_finish(const ImgOut& o, float vmax, int sat) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
    sat += __shfl_xor(sat, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    const unsigned bits = __float_as_uint(vmax);
    if (vmax > 0.f && bits > __hip_atomic_load(o.hdr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(o.hdr, bits);
    if (sat) atomicAdd(o.hdr + 3, (unsigned)sat);
  }
}

}  // namespace advoc
