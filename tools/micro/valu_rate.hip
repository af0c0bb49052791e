// Issue rate of vector instructions on gfx950, per SIMD: W waves per SIMD each run `iters` x 32 independent copies of one
// instruction; s_memtime around the loop (shader cycles).  Prints cycles per wave-instruction at the SIMD
// ( = elapsed cycles x 1 / (waves per SIMD x instructions per wave) ).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/micro/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ inline float sum_of(float v) { return v; }
__device__ inline float sum_of(f2 v) { return v.x + v.y; }

#define REP8(X) X X X X X X X X
#define BODY_T(NAME, ASM, T)                                                                                       \
  __global__ __launch_bounds__(256) void NAME(uint64_t* out, int iters, float seed) {                          \
    T a[8], b = T(seed), c = T(1.0f - seed);                                                                   \
    for (int i = 0; i < 8; ++i) a[i] = T(seed + i);                                                            \
    uint64_t t0 = __builtin_amdgcn_s_memtime();                                                                \
    for (int it = 0; it < iters; ++it) {                                                                       \
      for (int u = 0; u < 4; ++u) {                                                                            \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                                   \
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                     : "v"(b), "v"(c));                                                                        \
      }                                                                                                        \
    }                                                                                                          \
    uint64_t t1 = __builtin_amdgcn_s_memtime();                                                                \
    float s = 0; for (int i = 0; i < 8; ++i) s += sum_of(a[i]);                                                   \
    if (threadIdx.x % 64 == 0) out[(blockIdx.x * 256 + threadIdx.x) / 64] = t1 - t0;                          \
    if (s == 12345.678f) out[0] = 0;                                                                           \
  }
#define A_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define A_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define A_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_FMA2(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\nv_fma_f32 %" #i ", %" #i ", %9, %8\n"
#define A_ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define A_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define A_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define A_CND(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_CND64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
#define A_CND64V(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n"
#define A_CNDB(i) "v_cndmask_b32_e32 %" #i ", %8, %" #i ", vcc\n"
#define A_CMP(i) "v_cmp_lt_f32_e32 vcc, %" #i ", %8\n"
#define A_CMP64(i) "v_cmp_lt_f32_e64 s[20:21], %" #i ", %8\n"
#define A_CMPCND(i) "v_cmp_lt_f32_e32 vcc, %" #i ", %8\nv_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n"
#define A_MAX(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define A_SWAP(i) "v_swap_b32 %" #i ", %8\n"
#define A_PKADDSEL(i) "v_pk_add_f32 %" #i ", %" #i ", %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n"
BODY_T(k_pkfma, A_PKFMA, f2) BODY_T(k_pkadd, A_PKADD, f2) BODY_T(k_pkmul, A_PKMUL, f2) BODY_T(k_pkaddsel, A_PKADDSEL, f2)
BODY_T(k_fma, A_FMA, float) BODY_T(k_add, A_ADD, float) BODY_T(k_mov, A_MOV, float) BODY_T(k_sqrt, A_SQRT, float)
BODY_T(k_cnd, A_CND, float) BODY_T(k_fma2, A_FMA2, float) BODY_T(k_cnd64, A_CND64, float) BODY_T(k_cnd64v, A_CND64V, float) BODY_T(k_cndb, A_CNDB, float) BODY_T(k_cmp, A_CMP, float)
BODY_T(k_cmp64, A_CMP64, float) BODY_T(k_cmpcnd, A_CMPCND, float) BODY_T(k_max, A_MAX, float)

// LDS instructions: each wave works in its own 8 KB of LDS (conflict-free lane-linear addresses)
#define LBODY(NAME, ASM, T)                                                                                   \
  __global__ __launch_bounds__(256) void NAME(uint64_t* out, int iters, float seed) {                          \
    __shared__ float buf[4][2304];                                                                             \
    T a[8], b = T(seed);                                                                                       \
    for (int i = 0; i < 8; ++i) a[i] = T(seed + i);                                                            \
    for (int i = threadIdx.x; i < 4 * 2304; i += 256) (&buf[0][0])[i] = seed;                                  \
    __syncthreads();                                                                                           \
    unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&buf[threadIdx.x >> 6][0] + (threadIdx.x & 63) * sizeof(T); \
    int idx = ((threadIdx.x & 63) ^ 63) * 4;                                                                   \
    uint64_t t0 = __builtin_amdgcn_s_memtime();                                                                \
    for (int it = 0; it < iters; ++it) {                                                                       \
      for (int u = 0; u < 4; ++u) {                                                                            \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) "s_waitcnt lgkmcnt(0)\n"          \
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                     : "v"(b), "v"(addr), "v"(idx) : "memory");                                                \
      }                                                                                                        \
    }                                                                                                          \
    uint64_t t1 = __builtin_amdgcn_s_memtime();                                                                \
    float s = 0; for (int i = 0; i < 8; ++i) s += sum_of(a[i]);                                                \
    if (threadIdx.x % 64 == 0) out[(blockIdx.x * 256 + threadIdx.x) / 64] = t1 - t0;                          \
    if (s == 12345.678f) out[0] = 0;                                                                           \
  }
#define L_RD64(i) "ds_read_b64 %" #i ", %9 offset:" #i "*512\n"
#define L_RD32(i) "ds_read_b32 %" #i ", %9 offset:" #i "*256\n"
#define L_WR64(i) "ds_write_b64 %9, %" #i " offset:" #i "*512\n"
#define L_WR32(i) "ds_write_b32 %9, %" #i " offset:" #i "*256\n"
#define L_WR2B32(i) "ds_write2_b32 %9, %" #i ", %8 offset0:" #i "*2 offset1:" #i "*2+1\n"
#define L_RD2ST64(i) "ds_read2st64_b64 %" #i ", %9 offset0:" #i " offset1:" #i "+1\n"
#define L_BPERM(i) "ds_bpermute_b32 %" #i ", %10, %" #i "\n"
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ inline float sum_of(f4 v) { return v.x + v.y + v.z + v.w; }
LBODY(l_rd64, L_RD64, f2) LBODY(l_rd32, L_RD32, float) LBODY(l_wr64, L_WR64, f2) LBODY(l_wr32, L_WR32, float)
LBODY(l_bperm, L_BPERM, float)


int main() {
  uint64_t* d;
  const int wgs_per_cu[] = {1, 2, 4};
  hipMalloc(&d, 8 * 256 * 16 * 8);
  struct { const char* n; void (*k)(uint64_t*, int, float); int per; } ks[] = {
      {"v_pk_fma_f32", k_pkfma, 32}, {"v_pk_add_f32", k_pkadd, 32}, {"v_pk_mul_f32", k_pkmul, 32}, {"v_pk_add_f32 op_sel", k_pkaddsel, 32},
      {"v_fma_f32", k_fma, 32}, {"2 x v_fma_f32", k_fma2, 64}, {"v_add_f32", k_add, 32}, {"v_mov_b32", k_mov, 32}, {"v_sqrt_f32", k_sqrt, 32},
      {"v_cndmask_b32", k_cnd, 32}, {"v_cndmask_b32_e64 sgpr", k_cnd64, 32}, {"v_cndmask_b32_e64 vcc", k_cnd64v, 32}, {"v_cndmask_e32 (src swapped)", k_cndb, 32},
      {"v_cmp_lt_f32_e32 vcc", k_cmp, 32}, {"v_cmp_lt_f32_e64 sgpr", k_cmp64, 32}, {"v_cmp + v_cndmask vcc", k_cmpcnd, 64}, {"v_max_f32", k_max, 32},
      {"ds_read_b64", l_rd64, 32}, {"ds_read_b32", l_rd32, 32}, {"ds_write_b64", l_wr64, 32}, {"ds_write_b32", l_wr32, 32},
      {"ds_bpermute_b32", l_bperm, 32}};
  const int iters = 2000;
  for (auto& k : ks)
    for (int w : wgs_per_cu) {   // w workgroups of 4 waves per CU -> w waves per SIMD
      const int blocks = 256 * w;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, d, iters, 0.25f);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, d, iters, 0.25f);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      std::vector<uint64_t> h(blocks * 4);
      hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
      double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
      // s_memtime ticks at 100 MHz on some parts: report raw ticks too
      printf("%-22s waves/SIMD %d: %.0f ticks per wave (%.3f ms: %.2f ticks/ns), %.3f ticks = %.3f ns per wave-instruction at the SIMD\n", k.n, w, avg,
             ms, avg / (ms * 1e6), avg / ((double)iters * k.per * w), ms * 1e6 / ((double)iters * k.per * w));
    }
  return 0;
}
