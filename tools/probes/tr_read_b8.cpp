// Probe of ds_read_b64_tr_b8 (gfx950; __builtin_amdgcn_ds_read_tr8_b64_v2i32): LDS holds a [64 keys][64 dims] BYTE matrix with
// M[k][d] = the 16-bit id k * 64 + d split over two planes (low byte plane at 0, high byte plane at 4096), so that every byte a lane
// receives can be identified.  Each lane passes the address of 8 contiguous bytes: key = key0 + (i / 2), dims d0 + 8 * (i % 2) .. + 7 with
// i = lane & 15, d0 = 16 * ((lane >> 4) & 1), key0 = 8 * (lane >> 5) — the byte analogue of tools/probes/tr_read.cpp.  Prints, per lane, which
// (key, dim) each of its 8 result bytes came from.  Needed for an e4m3 P V product (V^T fragments out of a row-major V tile).
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/tr_read_b8.cpp -o tools/probes/bin/tr_read_b8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void probe(uint8_t* out) {
  __shared__ __attribute__((aligned(16))) uint8_t m[2 * 64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) { m[i] = (uint8_t)(i & 0xff); m[4096 + i] = (uint8_t)(i >> 8); }
  __syncthreads();
  const int lane = threadIdx.x, i = lane & 15;
  const int key = 8 * (lane >> 5) + (i >> 1), d = 16 * ((lane >> 4) & 1) + 8 * (i & 1);
  for (int plane = 0; plane < 2; ++plane) {
    v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(m + plane * 4096 + key * 64 + d));
    for (int j = 0; j < 8; ++j) out[(plane * 64 + lane) * 8 + j] = (uint8_t)(((unsigned)r[j >> 2] >> (8 * (j & 3))) & 0xff);
  }
}
int main() {
  uint8_t* d; hipMalloc(&d, 2 * 64 * 8);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  uint8_t h[2 * 64 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    const int i = l & 15;
    printf("lane %2d (addr k%2d,d%2d..):", l, 8 * (l >> 5) + (i >> 1), 16 * ((l >> 4) & 1) + 8 * (i & 1));
    for (int j = 0; j < 8; ++j) { const int id = h[l * 8 + j] | (h[(64 + l) * 8 + j] << 8); printf(" (k%2d,d%2d)", id / 64, id % 64); }
    printf("\n");
  }
  return 0;
}
