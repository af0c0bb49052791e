// Micro-benchmark: streaming WRITE bandwidth, 16 B / lane, and the thin-layer store pattern
// (a wave writes 8 consecutive 128-byte pixel rows per instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)

template <int U>
__global__ __launch_bounds__(256) void fill(float4* __restrict__ x, int64_t n4, float v) {
  const int64_t step = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const float4 f = make_float4(v, v, v, v);
  for (; i + (U - 1) * step < n4; i += U * step) {
#pragma unroll
    for (int u = 0; u < U; ++u) x[i + u * step] = f;
  }
}

// each block writes contiguous chunks (block-contiguous instead of grid-strided)
__global__ __launch_bounds__(256) void fill_chunk(float4* __restrict__ x, int64_t n4, float v) {
  const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
  const int64_t lo = blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
  const float4 f = make_float4(v, v, v, v);
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) x[i] = f;
}

// thin_fwd_kernel's pattern: a wave owns 16 pixels of 256 bytes; instruction b writes bytes [64 b, 64 b + 64) of each (lane
// (pixel, quarter) 16 bytes): 16 half lines per instruction, four instructions per 16 pixels
__global__ __launch_bounds__(256) void fill_pix64(float4* __restrict__ x, int64_t npix, float v) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pt = lane & 15, kg = lane >> 4;
  const float4 f = make_float4(v, v, v, v);
  const int64_t tiles = npix / 16;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < tiles; t += (int64_t)gridDim.x * 4) {
    float4* px = x + (t * 16 + pt) * 16;        // 256 bytes per pixel
#pragma unroll
    for (int b = 0; b < 4; ++b) px[4 * b + kg] = f;
  }
}
// the same bytes, a pixel's 256 bytes by 16 adjacent lanes: four pixels (1 KB contiguous) per instruction
__global__ __launch_bounds__(256) void fill_pix256(float4* __restrict__ x, int64_t npix, float v) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float4 f = make_float4(v, v, v, v);
  const int64_t tiles = npix / 16;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < tiles; t += (int64_t)gridDim.x * 4) {
#pragma unroll
    for (int b = 0; b < 4; ++b) x[(t * 16 + 4 * b) * 16 + lane] = f;
  }
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
  const size_t bytes = 268435456;
  float4* x; CK(hipMalloc(&x, bytes));
  auto rep = [&](const char* name, int blocks, float ms) { printf("%-28s blocks %5d %8.1f us  %7.1f GB/s\n", name, blocks, ms * 1e3, bytes / ms / 1e6); };
  for (int blocks : {256, 512, 1024, 2048, 4096, 16384}) {
    rep("fill U=1", blocks, timeit([&] { hipLaunchKernelGGL(fill<1>, dim3(blocks), dim3(256), 0, 0, x, (int64_t)(bytes / 16), 1.f); }));
    rep("fill U=4", blocks, timeit([&] { hipLaunchKernelGGL(fill<4>, dim3(blocks), dim3(256), 0, 0, x, (int64_t)(bytes / 16), 1.f); }));
    rep("fill chunked", blocks, timeit([&] { hipLaunchKernelGGL(fill_chunk, dim3(blocks), dim3(256), 0, 0, x, (int64_t)(bytes / 16), 1.f); }));
  }
  for (size_t mb : {256, 1024, 2048}) {
    float4* y; CK(hipMalloc(&y, mb << 20));
    const int64_t npix = (int64_t)(mb << 20) / 256;
    for (int blocks : {768, 1024, 2048}) {
      float ms = timeit([&] { hipLaunchKernelGGL(fill_pix64, dim3(blocks), dim3(256), 0, 0, y, npix, 1.f); });
      printf("fill_pix64  (16 x 64 B per instruction)  %4zu MB blocks %5d %8.1f us  %7.1f GB/s\n", mb, blocks, ms * 1e3, (double)(mb << 20) / ms / 1e6);
      ms = timeit([&] { hipLaunchKernelGGL(fill_pix256, dim3(blocks), dim3(256), 0, 0, y, npix, 1.f); });
      printf("fill_pix256 (1 KB contiguous per instr.) %4zu MB blocks %5d %8.1f us  %7.1f GB/s\n", mb, blocks, ms * 1e3, (double)(mb << 20) / ms / 1e6);
    }
    hipFree(y);
  }
  rep("hipMemsetAsync", 0, timeit([&] { hipMemsetAsync(x, 0, bytes, 0); }));
  return 0;
}
