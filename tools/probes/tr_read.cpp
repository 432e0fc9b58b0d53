// Probe of ds_read_b64_tr_b16 (gfx950): LDS holds a [64 keys][64 dims] bf16-sized matrix with M[k][d] = k * 64 + d (as
// uint16).  Each lane passes the address of 4 contiguous elements: key = key0 + (i / 4), dims d0 + 4 * (i % 4) .. + 3 with
// i = lane & 15, d0 = 16 * ((lane >> 4) & 1), key0 = 4 * (lane >> 5).  Prints what every lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t m[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) m[i] = (uint16_t)i;
  __syncthreads();
  const int lane = threadIdx.x, i = lane & 15;
  const int key = 4 * (lane >> 5) + (i >> 2), d = 16 * ((lane >> 4) & 1) + 4 * (i & 3);
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(m + key * 64 + d));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)r[j];
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) printf("  (k%2d,d%2d)", h[l * 4 + j] / 64, h[l * 4 + j] % 64);
    printf("\n");
  }
  return 0;
}
