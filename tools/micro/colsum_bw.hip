// Micro-benchmark: column sums of a [npix, c] fp32 matrix (bias-gradient access pattern), several
// streaming strategies.  hipcc --offload-arch=gfx950 -O3 -o colsum_bw colsum_bw.hip && ./colsum_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)

// A: grid-strided, U loads in flight (the shipped pattern)
template <int U>
__global__ __launch_bounds__(256) void strided(const float* __restrict__ dy, int64_t npix, int c, float* db) {
  __shared__ float4 red[256];
  const int quads = c / 4, groups = 256 / quads;
  const int tq = threadIdx.x % quads, tg = threadIdx.x / quads;
  float4 acc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) acc[u] = make_float4(0, 0, 0, 0);
  const int64_t step = (int64_t)gridDim.x * groups;
  int64_t pix = (int64_t)blockIdx.x * groups + tg;
  for (; pix + (U - 1) * step < npix; pix += U * step) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const float4*>(dy + (pix + u * step) * c + 4 * tq);
#pragma unroll
    for (int u = 0; u < U; ++u) { acc[u].x += v[u].x; acc[u].y += v[u].y; acc[u].z += v[u].z; acc[u].w += v[u].w; }
  }
  for (; pix < npix; pix += step) {
    float4 v = *reinterpret_cast<const float4*>(dy + pix * c + 4 * tq);
    acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
  }
  float4 s = acc[0];
#pragma unroll
  for (int u = 1; u < U; ++u) { s.x += acc[u].x; s.y += acc[u].y; s.z += acc[u].z; s.w += acc[u].w; }
  red[threadIdx.x] = s;
  __syncthreads();
  if (tg == 0) {
    float4 t = make_float4(0, 0, 0, 0);
    for (int k = 0; k < groups; ++k) { float4 r = red[k * quads + tq]; t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w; }
    unsafeAtomicAdd(db + 4 * tq, t.x); unsafeAtomicAdd(db + 4 * tq + 1, t.y); unsafeAtomicAdd(db + 4 * tq + 2, t.z); unsafeAtomicAdd(db + 4 * tq + 3, t.w);
  }
}

// B: each block owns a contiguous chunk; consecutive iterations read consecutive 4 KB * U
template <int U, bool NT>
__global__ __launch_bounds__(256) void chunked(const float* __restrict__ dy, int64_t npix, int c, float* db) {
  __shared__ float4 red[256];
  const int quads = c / 4, groups = 256 / quads;
  const int tq = threadIdx.x % quads, tg = threadIdx.x / quads;
  const int64_t per = (npix + gridDim.x - 1) / gridDim.x;
  const int64_t lo = blockIdx.x * per, hi = lo + per < npix ? lo + per : npix;
  float4 acc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) acc[u] = make_float4(0, 0, 0, 0);
  int64_t pix = lo + tg;
  for (; pix + (U - 1) * groups < hi; pix += U * groups) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float4* q = reinterpret_cast<const float4*>(dy + (pix + u * groups) * c + 4 * tq);
      if (NT) { v[u].x = __builtin_nontemporal_load(&q->x); v[u].y = __builtin_nontemporal_load(&q->y);
                v[u].z = __builtin_nontemporal_load(&q->z); v[u].w = __builtin_nontemporal_load(&q->w); }
      else v[u] = *q;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc[u].x += v[u].x; acc[u].y += v[u].y; acc[u].z += v[u].z; acc[u].w += v[u].w; }
  }
  for (; pix < hi; pix += groups) {
    float4 v = *reinterpret_cast<const float4*>(dy + pix * c + 4 * tq);
    acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
  }
  float4 s = acc[0];
#pragma unroll
  for (int u = 1; u < U; ++u) { s.x += acc[u].x; s.y += acc[u].y; s.z += acc[u].z; s.w += acc[u].w; }
  red[threadIdx.x] = s;
  __syncthreads();
  if (tg == 0) {
    float4 t = make_float4(0, 0, 0, 0);
    for (int k = 0; k < groups; ++k) { float4 r = red[k * quads + tq]; t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w; }
    unsafeAtomicAdd(db + 4 * tq, t.x); unsafeAtomicAdd(db + 4 * tq + 1, t.y); unsafeAtomicAdd(db + 4 * tq + 2, t.z); unsafeAtomicAdd(db + 4 * tq + 3, t.w);
  }
}

// C: pure streaming read, no column structure (upper bound): every thread sums float4s, grid-stride
template <int U>
__global__ __launch_bounds__(256) void stream(const float4* __restrict__ x, int64_t n4, float* out) {
  float4 acc = make_float4(0, 0, 0, 0);
  const int64_t step = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * step < n4; i += U * step) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = x[i + u * step];
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
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
  const int c = 32;
  const int64_t npix = 64LL * 128 * 256;       // D layer_1 output at batch 64: 268 MB
  const size_t bytes = sizeof(float) * npix * c;
  float *dy, *db;
  CK(hipMalloc(&dy, bytes)); CK(hipMalloc(&db, 4096));
  CK(hipMemset(dy, 0, bytes)); CK(hipMemset(db, 0, 4096));
  auto rep = [&](const char* name, float ms) { printf("%-34s %8.1f us  %7.1f GB/s\n", name, ms * 1e3, bytes / ms / 1e6); };
  for (int blocks : {256, 512, 768, 1024, 2048}) {
    printf("blocks %d\n", blocks);
    rep("strided U=4", timeit([&] { hipLaunchKernelGGL(strided<4>, dim3(blocks), dim3(256), 0, 0, dy, npix, c, db); }));
    rep("strided U=8", timeit([&] { hipLaunchKernelGGL(strided<8>, dim3(blocks), dim3(256), 0, 0, dy, npix, c, db); }));
    rep("chunked U=4", timeit([&] { hipLaunchKernelGGL((chunked<4, false>), dim3(blocks), dim3(256), 0, 0, dy, npix, c, db); }));
    rep("chunked U=8", timeit([&] { hipLaunchKernelGGL((chunked<8, false>), dim3(blocks), dim3(256), 0, 0, dy, npix, c, db); }));
    rep("chunked U=4 nontemporal", timeit([&] { hipLaunchKernelGGL((chunked<4, true>), dim3(blocks), dim3(256), 0, 0, dy, npix, c, db); }));
    rep("stream U=4", timeit([&] { hipLaunchKernelGGL(stream<4>, dim3(blocks), dim3(256), 0, 0, (const float4*)dy, (int64_t)(bytes / 16), db); }));
    rep("stream U=8", timeit([&] { hipLaunchKernelGGL(stream<8>, dim3(blocks), dim3(256), 0, 0, (const float4*)dy, (int64_t)(bytes / 16), db); }));
  }
  return 0;
}
