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
// offset is voffset + soffset into the descriptor.
//
// THE s_nop IS REQUIRED (r4).  On gfx9-generation parts, gfx950 included, an LDS-DMA instruction reads M0 one issue slot
// after a scalar write of M0 at the earliest: "SALU writes M0 -> LDS-DMA: 1 wait state" (LLVM inserts the s_nop itself
// behind its own M0 writes -- GCNHazardRecognizer, hasReadM0LdsDmaHazard; visible in the ISA of igemm_patch.hip, which uses
// the builtin -- but it does not look inside inline assembly).  Without it a DMA whose two instructions issue back to
// back goes to the LDS address of the wave's PREVIOUS DMA: one 1 KB block of a K tile lands in the wrong place and the
// intended block keeps stale bytes.  Whether they issue back to back depends on what else the SIMD has to issue: in r3
// the kernels of this header (per-tap gather, image weight gradient) gave wrong elements in ~1 of 3 runs ONLY while
// thin_wgrad_kernel's conflicted ds_add_f32 bursts ran beside them on the side stream, and never alone -- the "side-stream
// race" of NOTEBOOK.md section 5 (tools/micro/side_race_r4*.py: 8-10 of 12 runs wrong -> 0 of 60 with the s_nop).
// m0 is listed as clobbered so that the compiler never assumes a value of its own survives the statement.  (r5) clang
// answers every such clobber with "-Winline-asm: clobber list contains reserved registers: m0" -- 186 copies per
// translation unit, enough to hide a real warning; the clobber is what is wanted here (the compiler re-materialises M0
// for its own LDS-DMA builtins, ISA checked in igemm_patch.hip where both forms meet), so that ONE diagnostic is switched
// off around this ONE statement.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16(u32x4s rsrc, unsigned lds_base, int voffset, int soffset = 0) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_base), "v"(voffset),
               "s"(rsrc), "s"(soffset)
               : "memory", "m0");
}
#pragma clang diagnostic pop

// The rendezvous of an LDS-DMA ring: "my DMAs of the stage about to be read have landed (vmcnt <= VM), MY READS OF THE STAGE
// ABOUT TO BE OVERWRITTEN HAVE RETURNED (lgkmcnt 0), everybody is here".
//
// The lgkmcnt(0) is what r3 did not have, and it is the cause of the "side-stream race" of NOTEBOOK.md section 5 (found in r4,
// tools/micro/side_race_r4*.py).  s_barrier waits for no memory operation, and behind a raw __builtin_amdgcn_s_barrier() the
// compiler owes the LDS reads nothing either: it puts their s_waitcnt lgkmcnt in front of the first MFMA that uses the
// fragments, and it is free to sink that MFMA below the barrier -- in gather_gemm_h3_kernel<2,1,2,2> six ds_read_b128 of
// tile t were still outstanding when the wave passed the barrier of tile t + 1 (ISA, r3 build), behind which the OTHER
// waves issue the DMAs that overwrite exactly that stage.  A ds_write of another wave would queue behind those reads in the
// LDS instruction pipe; an LDS-DMA write arrives from the texture path and is ordered with nothing.  Normally the DMA's
// round trip (>= 500 cycles) is far longer than the reads stay queued, so the kernels were "safe by latency" -- until
// something jammed the LDS pipe of the CU: thin_wgrad_kernel's bias sums, 8 lanes per address on ds_add_f32, running on
// the side stream beside the backward-data launches (only the float LDS atomics did it: plain ds_write of the same
// addresses, integer atomics, or HBM-bound strangers on a second stream never reproduced it).  Then a wave's last fragment
// reads of a K tile returned the NEXT-BUT-ONE tile's bytes: one 32 x 32 accumulator block of one wave off by 1e-5 (low
// fp16 plane) to 1e-1 (high plane) relative, in ~70 % of the runs.  __syncthreads() has the wait built in (its fence drains
// lgkmcnt), which is why nobody meets this with ordinary barriers; the raw barrier is used here because __syncthreads()
// would also drain vmcnt to 0 where a ring keeps tiles in flight.
template <int VM>
__device__ __forceinline__ void dma_ring_barrier() {
  static_assert(VM >= 0 && VM < 64, "vmcnt is a 6-bit counter");
#ifdef ADVOC_RING_NO_LGKM      // (A/B builds only, tools/micro/lib_ab.sh: the r3 rendezvous, which has the race)
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(VM) : "memory");
#else
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VM) : "memory");
#endif
}

__device__ __forceinline__ unsigned lds_address(const void* generic_lds_pointer) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)generic_lds_pointer;
}

}  // namespace advoc
