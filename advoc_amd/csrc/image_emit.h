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

// (r5) power of two that puts an a-priori BOUND of the tensor's magnitude just below the top of the fp16 range -- at [2^15,
// 65000] when its mantissa allows, else at [2^14, 2^15): nothing can leave the range (the 1.0001 covers the rounding of the
// fp32 products the bound was computed with)
__device__ __forceinline__ float emit_up_scale_bounded(float bound) {
  const float bb = bound * 1.0001f;
  const unsigned b = __float_as_uint(bb);
  const int e = (int)((b >> 23) & 0xffu);
  if (e == 0 || e == 255) return 1.f;
  int s = 15 - (e - 127);
  s = s > 120 ? 120 : (s < -120 ? -120 : s);
  float up = __uint_as_float((unsigned)(s + 127) << 23);
  if (bb * up > 65000.f) up *= 0.5f;
  return up;
}

// sum_k |w| factor of the a-priori bound of a launch: the largest, over the launch's phases, of the sum over the phase's taps
// of the per-tap row-L1 maxima of the weight image (advoc_weight_images_l1_f32) -- or max|w| * taps * K without them
__device__ __forceinline__ float emit_weight_bound(const GatherGemmParams& p, int ktot) {
  const float crude = __uint_as_float(p.b_hdr[0]) * (float)(p.ntaps * ktot);
  if (!p.w_l1 || p.w_l1[2] == 0u || (int)p.w_l1[3] != ktot) return crude;
  float m = 0.f;
  for (int ph = 0; ph < p.nphase; ++ph) {
    float sum = 0.f;
    for (int i = 0; i < p.ntaps; ++i) {
      const int tp = p.tap[ph][i];
      if ((int)(int8_t)(tp & 0xff) == -128) continue;        // a padding tap of a short phase
      const unsigned wt = (unsigned)(tp >> 16);
      sum += wt < p.w_l1[2] ? __uint_as_float(p.w_l1[4 + wt]) : __uint_as_float(p.b_hdr[0]) * (float)ktot;
    }
    m = fmaxf(m, sum);
  }
  return fminf(m * 1.00001f, crude);
}

typedef float emit_f2 __attribute__((ext_vector_type(2)));
typedef _Float16 emit_h2 __attribute__((ext_vector_type(2)));

// Activation, scale and fp16 pair split of 4 values, two at a time on the packed fp32 / packed-convert instructions
// (v_pk_mul_f32, v_cvt_pk_f16_f32, v_pk_add_f32: ~7 VALU operations per value where the scalar form took 16 -- the
// emitting epilogues of the 64-channel layers had become a third slower than the plain ones).  `up` = the consumer's
// scale, or 0 for a row that must not count; vmaxs accumulates max |activation * up| BEFORE the clamp, so
// emit_finish can tell both the magnitude (vmaxs / up, exact: a power of two) and whether anything left the fp16 window.
// Value for value what pair_image_kernel<true> (image.hip) computes: RN(a), RN(a - RN(a)), clamp to +-65504.
__device__ __forceinline__ void emit_split4(float slope, float up, float4 v, float& vmaxs, uint2& h0, uint2& h1) {
  const emit_f2 t[2] = {{v.x, v.y}, {v.z, v.w}};
  unsigned lo[2], hi[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const emit_f2 st = t[k] * slope;
    // leaky / plain ReLU / identity (0 <= slope <= 1): the larger of t and slope t, as a median with +inf
    emit_f2 a = {__builtin_amdgcn_fmed3f(t[k].x, st.x, __builtin_inff()), __builtin_amdgcn_fmed3f(t[k].y, st.y, __builtin_inff())};
    a = a * up;
    vmaxs = fmaxf(vmaxs, fmaxf(fabsf(a.x), fabsf(a.y)));
    const emit_f2 c = {__builtin_amdgcn_fmed3f(a.x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(a.y, -65504.f, 65504.f)};
    const emit_h2 p0 = __builtin_convertvector(c, emit_h2);
    const emit_f2 r = c - __builtin_convertvector(p0, emit_f2);
    const emit_h2 p1 = __builtin_convertvector(r, emit_h2);
    lo[k] = __builtin_bit_cast(unsigned, p0);
    hi[k] = __builtin_bit_cast(unsigned, p1);
  }
  h0 = make_uint2(lo[0], lo[1]);
  h1 = make_uint2(hi[0], hi[1]);
}

// the 4 channels e .. e + 3 (e % 4 == 0) of one pixel, e = linear NHWC element index of the tensor
__device__ __forceinline__ void emit4(const ImgOut& o, float up, float4 v, unsigned e, float& vmaxs) {
  uint2 h0, h1;
  emit_split4(o.slope, up, v, vmaxs, h0, h1);
  uint16_t* dst = o.img + (size_t)(e >> 5) * 64 + (e & 31);
  *reinterpret_cast<uint2*>(dst) = h0;
  *reinterpret_cast<uint2*>(dst + 32) = h1;
}

// The same through a buffer descriptor, for epilogues that must not branch around their stores (igemm_patch.hip: the
// number of stores between a load and its use has to be known at compile time): byte_off = byte offset of element e in the
// fp32 tensor, or an out-of-range value for rows without a pixel -- the descriptor's range check drops those stores (and
// every store when the launch has no such consumer: num_records 0) and `valid` keeps them out of the statistics.
__device__ __forceinline__ void emit4_buffer(__amdgpu_buffer_rsrc_t rs, float slope, float up, float4 v, unsigned byte_off,
                                             bool valid, float& vmaxs) {
  uint2 h0, h1;
  emit_split4(slope, valid ? up : 0.f, v, vmaxs, h0, h1);
  // element e = byte_off / 4 -> halves (e >> 5) * 64 + (e & 31) -> bytes: ((e >> 5) * 64 + (e & 31)) * 2
  const unsigned e = byte_off >> 2;
  const unsigned ib = byte_off >= 0xffffff00u ? 0xffffff00u : (((e >> 5) << 6) + (e & 31)) * 2u;
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  __builtin_amdgcn_raw_buffer_store_b64((u32x2){h0.x, h0.y}, rs, ib, 0, 0);
  __builtin_amdgcn_raw_buffer_store_b64((u32x2){h1.x, h1.y}, rs, ib, 64, 0);
}

// once per wave, after its last emit4: the largest magnitude it saw and its out-of-window count into the header (the
// count is in LANES that saw a value beyond the window, not in values: refit_image_kernel only asks whether it moved)
__device__ __forceinline__ void emit_finish(const ImgOut& o, float up, float vmaxs) {
  int sat = vmaxs > 65504.f ? 1 : 0;
  float vmax = vmaxs / up;
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
