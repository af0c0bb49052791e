// Hardware probe (gfx950), round 4: does a co-resident kernel's LDS float atomic (ds_add_f32) disturb ANOTHER kernel's
// LDS-DMA (buffer_load_dwordx4 ... lds) on the same compute unit?
//
// Victim `dma_check`: the loop shape of the image kernels (igemm_h3.hip / igemm_patch.hip / wgrad_h3.hip) --
//   s_waitcnt vmcnt(0); s_barrier; issue the NEXT stage's DMAs; ds_read the CURRENT stage -- on a source buffer whose
//   32-bit word i holds i, so every value read from LDS can be checked and a wrong one classified (the value the same
//   LDS slot held two iterations ago = stale, anything else = foreign).
// Culprit `lds_hammer<MODE>` on a second stream: 0 ds_add_f32 on 32 conflicted addresses, 1 plain ds_write of the same
//   addresses, 2 ds_add_u32, 3 ds_add_rtn_f32 (value used), 4 nothing in LDS (global traffic only).
//
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_vs_dma tools/micro/lds_atomic_vs_dma.hip && ./lds_atomic_vs_dma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(u32x4s rsrc, unsigned lds_base, int voffset, int soffset) {
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_base), "v"(voffset), "s"(rsrc),
               "s"(soffset)
               : "memory");
}

constexpr int kStageBytes = 16384;           // 4 waves x 4 DMAs x 1 KB
constexpr int kWords = kStageBytes / 4;

// stats[0] mismatching words, [1] of them stale (= the slot's content two iterations ago), [2] zero, [3..] first samples
__global__ __launch_bounds__(256) void dma_check(const unsigned* __restrict__ src, unsigned src_bytes, int iters,
                                                 unsigned* __restrict__ stats, int extra_lgkm_wait) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned a = (unsigned)(uintptr_t)src;
  const u32x4s rs = {(unsigned)(uintptr_t)src, (unsigned)((uintptr_t)src >> 32) & 0xffffu, src_bytes, 0x00020000u};
  (void)a;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  // iteration it, block blockIdx.x: stage words [0, kWords) = source words base(it) + [0, kWords)
  const unsigned span = src_bytes / 4 - kWords;
#define BASE_OF(IT) ((unsigned)(((unsigned)blockIdx.x * 7919u + (unsigned)(IT) * 104729u) % span) & ~3u)
#define ISSUE(IT)                                                                                       \
  {                                                                                                     \
    const unsigned b_ = BASE_OF(IT);                                                                    \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                     \
      const int chunk = wave * 4 + j;                                                                   \
      dma16(rs, lds0 + ((IT) & 1) * kStageBytes + chunk * 1024, (int)(b_ * 4 + chunk * 1024 + lane * 16), 0); \
    }                                                                                                   \
  }
  unsigned bad = 0, stale = 0, zero = 0;
  ISSUE(0);
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (extra_lgkm_wait) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (it + 1 < iters) ISSUE(it + 1);
    const unsigned b = BASE_OF(it), bold = it >= 2 ? BASE_OF(it - 2) : 0xffffffffu;
    // every thread checks 16 words of ANOTHER wave's chunks
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int w = ((tid + 64 + r * 1024 / 4 * 1) * 4) % kWords;      // word index inside the stage, 16-byte aligned
      const uint4 v = *reinterpret_cast<const uint4*>(smem + (it & 1) * kStageBytes + w * 4);
      const unsigned got[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned want = b + w + k;
        if (got[k] != want) {
          ++bad;
          if (got[k] == bold + w + k) ++stale;
          if (got[k] == 0) ++zero;
          const unsigned slot = atomicAdd(stats + 3, 1u);
          if (slot < 16) { stats[8 + 4 * slot] = want; stats[9 + 4 * slot] = got[k]; stats[10 + 4 * slot] = it; stats[11 + 4 * slot] = blockIdx.x; }
        }
      }
    }
  }
  if (bad) { atomicAdd(stats, bad); atomicAdd(stats + 1, stale); atomicAdd(stats + 2, zero); }
}

template <int MODE>
__global__ __launch_bounds__(256) void lds_hammer(float* __restrict__ out, int reps) {
  __shared__ float s[256];
  const int tid = threadIdx.x;
  s[tid] = 0.f;
  __syncthreads();
  float acc = 0.f;
  float* d = s + (tid & 7) * 4;                 // 8 lanes per address: conflicted, like thin_wgrad_kernel's bias sums
  for (int r = 0; r < reps; ++r) {
    const float v = 1e-9f * (float)(tid + r);
    if (MODE == 0) { atomicAdd(d, v); atomicAdd(d + 1, v); atomicAdd(d + 2, v); atomicAdd(d + 3, v); }
    if (MODE == 1) { d[0] = v; d[1] = v; d[2] = v; d[3] = v; }
    if (MODE == 2) {
      unsigned* u = reinterpret_cast<unsigned*>(d);
      atomicAdd(u, 1u); atomicAdd(u + 1, 1u); atomicAdd(u + 2, 1u); atomicAdd(u + 3, 1u);
    }
    if (MODE == 3) { acc += atomicAdd(d, v) + atomicAdd(d + 1, v) + atomicAdd(d + 2, v) + atomicAdd(d + 3, v); }
    if (MODE == 4) { acc += out[(blockIdx.x * 256 + tid + r * 65536) & 0xfffff]; }
    __syncthreads();
  }
  out[blockIdx.x * 256 + tid] = s[tid] + acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  const unsigned src_bytes = 64u << 20;
  std::vector<unsigned> h(src_bytes / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)i;
  unsigned *src, *stats;
  float* out;
  CK(hipMalloc(&src, src_bytes));
  CK(hipMalloc(&stats, 4096));
  CK(hipMalloc(&out, 4 << 20));
  CK(hipMemset(out, 0, 4 << 20));
  CK(hipMemcpy(src, h.data(), src_bytes, hipMemcpyHostToDevice));
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa));
  CK(hipStreamCreate(&sb));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dma_check), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStageBytes));
  const char* names[] = {"ds_add_f32 (conflicted)", "plain ds_write", "ds_add_u32", "ds_add_rtn_f32", "no LDS traffic (global loads)", "victim alone"};
  for (int extra = 0; extra < 2; ++extra)
    for (int mode = 5; mode >= 0; --mode) {
      unsigned tot[3] = {0, 0, 0};
      unsigned sample[64] = {0};
      for (int trial = 0; trial < 5; ++trial) {
        CK(hipMemset(stats, 0, 4096));
        CK(hipDeviceSynchronize());
        if (mode < 5) {
          for (int l = 0; l < 12; ++l) {
            switch (mode) {
              case 0: hipLaunchKernelGGL(lds_hammer<0>, dim3(1024), dim3(256), 0, sb, out, 400); break;
              case 1: hipLaunchKernelGGL(lds_hammer<1>, dim3(1024), dim3(256), 0, sb, out, 400); break;
              case 2: hipLaunchKernelGGL(lds_hammer<2>, dim3(1024), dim3(256), 0, sb, out, 400); break;
              case 3: hipLaunchKernelGGL(lds_hammer<3>, dim3(1024), dim3(256), 0, sb, out, 400); break;
              default: hipLaunchKernelGGL(lds_hammer<4>, dim3(1024), dim3(256), 0, sb, out, 400); break;
            }
          }
        }
        for (int l = 0; l < 4; ++l)
          hipLaunchKernelGGL(dma_check, dim3(512), dim3(256), 2 * kStageBytes, sa, src, src_bytes, iters, stats, extra);
        CK(hipDeviceSynchronize());
        unsigned st[72];
        CK(hipMemcpy(st, stats, sizeof(st), hipMemcpyDeviceToHost));
        for (int k = 0; k < 3; ++k) tot[k] += st[k];
        if (st[0] && !sample[0]) for (int k = 0; k < 64; ++k) sample[k] = st[8 + k];
      }
      printf("%s beside: %-32s wrong words %10u  (stale %u, zero %u)\n", extra ? "[+lgkmcnt(0)] " : "", names[mode], tot[0], tot[1], tot[2]);
      if (tot[0])
        for (int k = 0; k < 4; ++k)
          printf("     sample: want %08x got %08x (as float %g) iteration %u block %u\n", sample[4 * k], sample[4 * k + 1],
                 *reinterpret_cast<float*>(&sample[4 * k + 1]), sample[4 * k + 2], sample[4 * k + 3]);
      fflush(stdout);
    }
  return 0;
}
