// Micro-benchmark: the fp16 pair image pass (image.hip, pair_image_kernel<true,false>) against a float4 copy of the same
// bytes, with variants of the loop (loads in flight, non-temporal hints, block-contiguous chunks).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void copy4(const float4* __restrict__ x, float4* __restrict__ y, int64_t n4) {
  const int64_t step = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += step) y[i] = x[i];
}

__device__ __forceinline__ void emit(const float v[8], float up, __half* img, int64_t e, bool nt) {
  __half2 h0[4], h1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = v[2 * j] * up, b = v[2 * j + 1] * up;
    const __half a0 = __float2half_rn(a), b0 = __float2half_rn(b);
    h0[j] = __halves2half2(a0, b0);
    h1[j] = __halves2half2(__float2half_rn(a - __half2float(a0)), __float2half_rn(b - __half2float(b0)));
  }
  __half* o = img + (e >> 5) * 64 + ((e >> 3) & 3) * 8;
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  if (nt) {
    __builtin_nontemporal_store(*reinterpret_cast<const u4*>(h0), reinterpret_cast<u4*>(o));
    __builtin_nontemporal_store(*reinterpret_cast<const u4*>(h1), reinterpret_cast<u4*>(o + 32));
  } else {
    *reinterpret_cast<u4*>(o) = *reinterpret_cast<const u4*>(h0);
    *reinterpret_cast<u4*>(o + 32) = *reinterpret_cast<const u4*>(h1);
  }
}

// U iterations' loads issued before any arithmetic; NT: non-temporal loads and stores; ACT: leaky relu + amax + saturation
template <int U, bool NT, bool CHUNK, int MODE>
__global__ __launch_bounds__(256) void image(const float* __restrict__ x, __half* __restrict__ img, int64_t n8, float up,
                                             float slope, unsigned* __restrict__ hdr) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  int64_t step = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, end = n8;
  if (CHUNK) {
    const int64_t per = ((n8 + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    i = blockIdx.x * per + threadIdx.x;
    end = (blockIdx.x + 1) * per < n8 ? (blockIdx.x + 1) * per : n8;
    step = 256;
  }
  float vmax = 0.f;
  int sat = 0;
  for (; i < end; i += U * step) {
    f4 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t k = i + u * step;
      if (k < end) {
        const f4* p = reinterpret_cast<const f4*>(x + k * 8);
        if (NT) { a[u] = __builtin_nontemporal_load(p); b[u] = __builtin_nontemporal_load(p + 1); }
        else { a[u] = p[0]; b[u] = p[1]; }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t k = i + u * step;
      if (k < end) {
        float v[8] = {a[u].x, a[u].y, a[u].z, a[u].w, b[u].x, b[u].y, b[u].z, b[u].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[j] = fmaxf(v[j], slope * v[j]);
          vmax = fmaxf(vmax, fabsf(v[j]));
          const float t = v[j] * up;
          if (fabsf(t) > 65504.f) { v[j] = copysignf(65504.f, t) / up; ++sat; }
        }
        emit(v, up, img, k * 8, NT);
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
  if (MODE == 0) {            // one atomic per wave
    if ((threadIdx.x & 63) == 0 && vmax > 0.f) atomicMax(hdr, __float_as_uint(vmax));
  } else {
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      vmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
      // MODE 1: one atomic per block; MODE 2: only when the value seen in memory is smaller (stale reads are smaller)
      if (MODE == 1 ? vmax > 0.f : __float_as_uint(vmax) > __hip_atomic_load(hdr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(hdr, __float_as_uint(vmax));
    }
  }
  if (sat) atomicAdd(hdr + 3, (unsigned)sat);
}

template <typename F>
float timeit(F f, int reps = 10) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); f();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  for (int64_t elems : {(int64_t)128 * 128 * 256 * 64, (int64_t)64 * 64 * 129 * 128, (int64_t)64 * 16 * 33 * 512, (int64_t)64 * 4 * 9 * 512}) {
    const size_t bytes = elems * 4;
    float* x; __half* img; unsigned* hdr;
    CK(hipMalloc(&x, bytes)); CK(hipMalloc(&img, bytes)); CK(hipMalloc(&hdr, 16));
    { std::vector<float> h(elems); unsigned r = 1; for (auto& v : h) { r = r * 1664525u + 1013904223u; v = ((int)(r >> 8) - (1 << 23)) * (1.f / (1 << 23)); } CK(hipMemcpy(x, h.data(), bytes, hipMemcpyHostToDevice)); } CK(hipMemset(hdr, 0, 16));
    const int64_t n8 = elems / 8;
    printf("elems %lld (%.0f MB in, same out)\n", (long long)elems, bytes / 1e6);
    auto rep = [&](const char* name, int blocks, float ms) {
      printf("  %-34s blocks %6d %8.1f us  %7.1f GB/s\n", name, blocks, ms * 1e3, 2.0 * bytes / ms / 1e6);
    };
    for (int blocks : {512, 1024, 2048, 4096}) {
      rep("copy4", blocks, timeit([&] { hipLaunchKernelGGL(copy4, dim3(blocks), dim3(256), 0, 0, (const float4*)x, (float4*)img, (int64_t)(bytes / 16)); }));
#define RUN(U, NT, CH, M) rep("image U=" #U " NT=" #NT " CHUNK=" #CH " MODE=" #M, blocks, timeit([&] { \
        hipMemsetAsync(hdr, 0, 4, 0); \
        hipLaunchKernelGGL((image<U, NT, CH, M>), dim3(blocks), dim3(256), 0, 0, x, img, n8, 512.f, 0.2f, hdr); }))
      RUN(1, false, false, 1);
      RUN(1, false, false, 2);
      RUN(2, false, false, 2);
      RUN(4, false, false, 2);
      RUN(2, false, true, 2);
      RUN(4, true, false, 2);
    }
    CK(hipFree(x)); CK(hipFree(img)); CK(hipFree(hdr));
  }
  return 0;
}
