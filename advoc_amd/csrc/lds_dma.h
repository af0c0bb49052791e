// LDS-DMA (buffer_load ... lds) as inline assembly, on purpose.  Through __builtin_amdgcn_raw_ptr_buffer_load_lds the
// compiler knows the instruction writes LDS, cannot tell one stage of a ring from another, and therefore puts
// `s_waitcnt vmcnt(0)` in front of the first ds_read that follows in program order: in a loop of the shape
//     wait; barrier; issue the NEXT tile's loads; ds_read the CURRENT tile; MFMA
// the reads of the current stage waited for the loads just issued, every K tile -- nothing was ever in flight under the
// MFMAs (r3: found in the ISA of wgrad_h3_256_kernel, 0.845 -> 0.783 ms).  The compiler does not track inline assembly
// in its wait counters, so the hand-placed `s_waitcnt vmcnt(N)` in front of each barrier is the only wait these loads get
// (and the only one they need); loads the compiler does track only ever wait more than necessary because of them.
#pragma once
#include <stdint.h>

namespace advoc {

typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

// raw buffer descriptor: base, num_records = bytes, the 0x00020000 flags of the builtin descriptors (offsets at or beyond
// `bytes` -- 0x80000000 for "no pixel" -- read as zeros)
__device__ __forceinline__ u32x4s dma_rsrc(const void* base, unsigned bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  return (u32x4s){(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}

// 16 bytes per lane: lane l's bytes land at LDS byte address lds_base + 16 l (lds_base wave-uniform); the global byte
// offset is voffset + soffset into the descriptor.  (m0 is a reserved register: the compiler keeps nothing in it.)
__device__ __forceinline__ void dma16(u32x4s rsrc, unsigned lds_base, int voffset, int soffset = 0) {
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_base), "v"(voffset), "s"(rsrc),
               "s"(soffset)
               : "memory");
}

__device__ __forceinline__ unsigned lds_address(const void* generic_lds_pointer) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)generic_lds_pointer;
}

}  // namespace advoc
