// Shared helpers for the gfx950 kernels of libadvoc_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "advoc_hip.h"

// hipGetLastError() is sticky per host thread and the host framework leaves benign codes
// there (e.g. hipErrorNotReady from event queries), so clear it right before each launch and
// read it right after: only OUR launch's status is reported.
#define ADVOC_CLEAR_LAUNCH_ERROR() (void)hipGetLastError()
#define ADVOC_RETURN_IF_LAUNCH_FAILED()                 \
  do {                                                  \
    const hipError_t advoc_e_ = hipGetLastError();      \
    if (advoc_e_ != hipSuccess) {                       \
      advoc::note_hip_error(advoc_e_);                  \
      return ADVOC_ERR_HIP;                             \
    }                                                   \
  } while (0)

namespace advoc {

// last HIP error seen by this host thread inside the library (diagnostics only)
void note_hip_error(hipError_t e);

constexpr int kWave = 64;  // CDNA wavefront

static inline hipStream_t as_stream(advoc_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Compiler + LDS ordering point for data exchanged between lanes of ONE wavefront through
// LDS (the LDS pipeline itself is in-order per wave, so no s_barrier is needed).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

}  // namespace advoc
