// Probe: the cross-lane sum helpers of csrc/common.h (DPP row ops + gfx950 permlane swaps) against the obvious sums.
#include "../../udifftext_amd/csrc/common.h"
#include <stdio.h>
__global__ void k(float* out) {
  const int lane = threadIdx.x;
  const float v = (float)(1 + lane * 3 % 17) + 0.25f * lane;
  out[lane] = row16_sum(v);
  out[64 + lane] = xor16_sum(row16_sum(v));
  out[128 + lane] = xor32_sum(xor16_sum(dpp_add<0x128>(v)));
  out[192 + lane] = v;
}
int main() {
  float* d; float h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    float r16 = 0, r32 = 0, rows = 0;
    for (int j = 0; j < 64; ++j) {
      if ((j >> 4) == (l >> 4)) r16 += h[192 + j];
      if ((j >> 5) == (l >> 5)) r32 += h[192 + j];
      if ((j & 7) == (l & 7)) rows += h[192 + j];
    }
    if (fabsf(h[l] - r16) > 1e-3f || fabsf(h[64 + l] - r32) > 1e-3f || fabsf(h[128 + l] - rows) > 1e-3f) {
      if (bad < 8) printf("lane %d: row16 %g (want %g) half32 %g (want %g) rowlanes %g (want %g)\n", l, h[l], r16, h[64 + l], r32, h[128 + l], rows);
      ++bad;
    }
  }
  printf("dpp sums: %s (%d bad lanes)\n", bad ? "MISMATCH" : "ok", bad);
  return 0;
}
