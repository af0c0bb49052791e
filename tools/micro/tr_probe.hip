// Semantics probe of ds_read_b64_tr_b16 on gfx950: LDS holds R[k][n] = 100 k + n (row stride 64 halves); every 16-lane
// group reads one [4 k][16 n] block, lane l of the group at &R[k0 + l / 4][n0 + 4 (l % 4)]; prints what each lane got.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short short4v __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)((i / 64) * 100 + (i % 64));
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, ll = l & 15;
  const int k0 = 4 * (g >> 1), n0 = 16 * (g & 1);      // groups 0,1: k 0-3, n 0-15 / 16-31; groups 2,3: k 4-7
  const short* a = lds + (k0 + ll / 4) * 64 + n0 + 4 * (ll % 4);
  short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)a);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  return 0;
}
