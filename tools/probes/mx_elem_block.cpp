// Probe: which E8M0 scale (the one supplied by half-wave 0 or by half-wave 1) multiplies byte t of lane L's operand registers in
// v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3)?  All elements 1.0 except ONE element = 2.0; the scaled operand's half-wave-0 lanes carry
// scale 2^1, its half-wave-1 lanes 2^0; the other operand has unit scales.  C[row][0] = 96 + 2 if the element sits in the block scaled
// by half-wave 0, 96 + 1 if in the block scaled by half-wave 1.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/mx_elem_block.cpp -o tools/probes/bin/mx_elem
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void probe(int L, int t, int which, float* C) {
  const int lane = threadIdx.x;
  const int row = lane & 31, hi = lane >> 5;
  uint8_t av[32], bv[32];
  for (int j = 0; j < 32; ++j) av[j] = bv[j] = 0x38;
  if (lane == L) { if (which == 0) av[t] = 0x40; else bv[t] = 0x40; }
  i32x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = av[4 * j] | (av[4 * j + 1] << 8) | (av[4 * j + 2] << 16) | (av[4 * j + 3] << 24);
    b[j] = bv[4 * j] | (bv[4 * j + 1] << 8) | (bv[4 * j + 2] << 16) | (bv[4 * j + 3] << 24);
  }
  const int s = hi == 0 ? 128 : 127;
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, which == 0 ? s : 127, 0, which == 1 ? s : 127);
  for (int r = 0; r < 16; ++r) {
    const int m = (r & 3) + 8 * (r >> 2) + 4 * hi;
    C[m * 32 + row] = c[r];
  }
}

int main() {
  float* dC; hipMalloc(&dC, 32 * 32 * 4);
  for (int which = 0; which < 2; ++which)
    for (int L = 0; L < 64; L += 32) {
      printf("%c operand, lane %2d: byte t -> scaled by half-wave: ", which ? 'B' : 'A', L);
      for (int t = 0; t < 32; ++t) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, L, t, which, dC);
        float out[32 * 32];
        hipMemcpy(out, dC, sizeof(out), hipMemcpyDeviceToHost);
        const float v = out[0];            // row 0 (A) / column 0 (B) holds lane 0's and lane 32's row
        printf("%c", v == 98.f ? '0' : v == 97.f ? '1' : '?');
      }
      printf("\n");
    }
  return 0;
}
