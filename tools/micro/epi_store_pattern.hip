// Micro-benchmark: the store pattern of a patch-kernel epilogue against alternatives, one 512-thread workgroup per CU
// (persistent, 140 KB of LDS claimed so that nothing else fits), no arithmetic: which address pattern does the chip's write
// path take at what rate?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/epi_store_pattern.hip -o /tmp/esp && /tmp/esp
// Tensor: [pixels][C] fp32, a tile = 16 x 16 pixels of an image row pitch W (NHWC), C = 128 (one column tile of the 128-channel
// instances).  Waves: (point quarter wave >> 1, column half wave & 1), as patch_gemm_h3_kernel<3,.>.
//   MODE 0  as the kernels do today: a store instruction = 8 pixels x 128 B (one 32-channel block), the pixel's other blocks
//           by later instructions / the other wave
//   MODE 1  8 store instructions per 32 pixels, each 4 pixels x 256 B (the wave's whole column half of a pixel at once)
//   MODE 2  waves own whole pixels: each instruction 2 pixels x 512 B
//   MODE 3  linear: each instruction 1 KB contiguous (what a fill does), same bytes per workgroup
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)

constexpr int C = 128;

template <int MODE>
__global__ __launch_bounds__(512) void epi(float4* __restrict__ y, int tiles, int tiles_x, int W, int H, float v) {
  extern __shared__ float smem[];
  if (v == 123.f) smem[threadIdx.x] = v;            // (keeps the LDS claim alive)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float4 f = make_float4(v, v, v, v);
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int per_img = tiles_x * (H / 16);
    const int img = tile / per_img, t = tile % per_img;
    const int y0 = (t / tiles_x) * 16, x0 = (t % tiles_x) * 16;
    const long base = ((long)img * H + y0) * W + x0;            // pixel index of the tile's corner
    if (MODE == 0) {
      const int wm = wave >> 1, wn = wave & 1, trow = lane >> 3, tq = lane & 7;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            const int r = trow + 8 * ps;                         // point of the 32-point block: 2 patch rows x 16
            const long pix = base + (long)(wm * 4 + i * 2 + (r >> 4)) * W + (r & 15);
            y[pix * (C / 4) + wn * 16 + j * 8 + tq] = f;
          }
    } else if (MODE == 1) {
      const int wm = wave >> 1, wn = wave & 1, trow = lane >> 4, tq = lane & 15;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
          const int r = trow + 4 * ps;
          const long pix = base + (long)(wm * 4 + i * 2 + (r >> 4)) * W + (r & 15);
          y[pix * (C / 4) + wn * 16 + tq] = f;
        }
    } else if (MODE == 2) {
      const int trow = lane >> 5, tq = lane & 31;                 // wave: 32 pixels (2 patch rows), all 128 channels
#pragma unroll
      for (int ps = 0; ps < 16; ++ps) {
        const int r = trow + 2 * ps;
        const long pix = base + (long)(wave * 2 + (r >> 4)) * W + (r & 15);
        y[pix * (C / 4) + tq] = f;
      }
    } else {
      // the tile's 256 x 128 floats as one linear 128 KB chunk (wrong addresses for a real tensor: the rate of the bytes)
#pragma unroll
      for (int k = 0; k < 16; ++k) y[(long)tile * 8192 + (wave * 16 + k) * 64 + lane] = f;
    }
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
  const int B = 64, H = 64, W = 128;                       // 64 x 64 x 128 x 128 fp32 = 268 MB
  const size_t bytes = (size_t)B * H * W * C * 4;
  float4* y; CK(hipMalloc(&y, bytes + (1 << 20)));
  const int tiles_x = W / 16, tiles = B * (H / 16) * tiles_x;
  const int lds = 140 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(epi<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(epi<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(epi<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(epi<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  for (int wgs : {256, 128, 64}) {
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL(epi<0>, dim3(wgs), dim3(512), lds, 0, y, tiles, tiles_x, W, H, 1.f); });
    printf("wgs %3d  8 px x 128 B per instruction (today)   %8.1f us %7.1f GB/s  %6.1f KB/us per CU\n", wgs, ms * 1e3, bytes / ms / 1e6, bytes / ms / 1e6 / wgs);
    ms = timeit([&] { hipLaunchKernelGGL(epi<1>, dim3(wgs), dim3(512), lds, 0, y, tiles, tiles_x, W, H, 1.f); });
    printf("wgs %3d  4 px x 256 B per instruction           %8.1f us %7.1f GB/s  %6.1f KB/us per CU\n", wgs, ms * 1e3, bytes / ms / 1e6, bytes / ms / 1e6 / wgs);
    ms = timeit([&] { hipLaunchKernelGGL(epi<2>, dim3(wgs), dim3(512), lds, 0, y, tiles, tiles_x, W, H, 1.f); });
    printf("wgs %3d  2 px x 512 B per instruction           %8.1f us %7.1f GB/s  %6.1f KB/us per CU\n", wgs, ms * 1e3, bytes / ms / 1e6, bytes / ms / 1e6 / wgs);
    ms = timeit([&] { hipLaunchKernelGGL(epi<3>, dim3(wgs), dim3(512), lds, 0, y, tiles, tiles_x, W, H, 1.f); });
    printf("wgs %3d  1 KB contiguous per instruction        %8.1f us %7.1f GB/s  %6.1f KB/us per CU\n", wgs, ms * 1e3, bytes / ms / 1e6, bytes / ms / 1e6 / wgs);
  }
  return 0;
}
