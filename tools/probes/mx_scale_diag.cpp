// Diagnostic probe for the block scales of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 A and B, all elements 1.0):
// C[i][j] = sum over the two 32-element k-blocks of 32 * 2^(sa(i, kb) - 127) * 2^(sb(j, kb) - 127).
// One lane L at a time gets scale 2^1 (E8M0 128) in byte `byte` of its A-scale (or B-scale) VGPR, everything else 127; the
// kernel runs with opsel = `sel`.  Prints which outputs moved, for a few L / byte / sel combinations.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/mx_scale_diag.cpp -o tools/probes/bin/mx_diag
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SEL>
__global__ void probe(const uint32_t* SA, const uint32_t* SB, float* C) {
  const int lane = threadIdx.x;
  const int row = lane & 31, hi = lane >> 5;
  i32x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = 0x38383838; b[j] = 0x38383838; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, SEL, (int)SA[lane], SEL, (int)SB[lane]);
  for (int r = 0; r < 16; ++r) {
    const int m = (r & 3) + 8 * (r >> 2) + 4 * hi;
    C[m * 32 + row] = c[r];
  }
}

static void run(int sel, const uint32_t* dSA, const uint32_t* dSB, float* dC) {
  if (sel == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, dSA, dSB, dC);
  if (sel == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, dSA, dSB, dC);
  if (sel == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, dSA, dSB, dC);
  if (sel == 3) hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64), 0, 0, dSA, dSB, dC);
}

int main() {
  uint32_t *dSA, *dSB; float* dC;
  hipMalloc(&dSA, 64 * 4); hipMalloc(&dSB, 64 * 4); hipMalloc(&dC, 32 * 32 * 4);
  const int lanes[] = {0, 5, 31, 32, 37, 63};
  for (int which = 0; which < 2; ++which)          // 0: perturb the A scale, 1: the B scale
    for (int byte = 0; byte < 4; ++byte)
      for (int sel = 0; sel < 4; ++sel)
        for (int li = 0; li < 6; ++li) {
          const int L = lanes[li];
          uint32_t sa[64], sb[64];
          for (int i = 0; i < 64; ++i) sa[i] = sb[i] = 0x7f7f7f7fu;
          uint32_t* t = which ? sb : sa;
          t[L] = (t[L] & ~(0xffu << (8 * byte))) | (128u << (8 * byte));
          hipMemcpy(dSA, sa, sizeof(sa), hipMemcpyHostToDevice); hipMemcpy(dSB, sb, sizeof(sb), hipMemcpyHostToDevice);
          run(sel, dSA, dSB, dC);
          float out[32 * 32];
          hipMemcpy(out, dC, sizeof(out), hipMemcpyDeviceToHost);
          // summarise: set of rows / columns whose value differs from 64, and the values seen
          int rmin = 99, rmax = -1, cmin = 99, cmax = -1, n = 0; float v0 = 0;
          for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j)
              if (out[i * 32 + j] != 64.f) {
                if (!n) v0 = out[i * 32 + j];
                ++n; if (i < rmin) rmin = i; if (i > rmax) rmax = i; if (j < cmin) cmin = j; if (j > cmax) cmax = j;
              }
          if (li == 0 || n)
            printf("%c-scale byte %d of lane %2d = 128, opsel %d: %4d outputs != 64 (rows %d..%d, cols %d..%d, first value %.1f; C[0][0] = %.1f)\n",
                   which ? 'B' : 'A', byte, L, sel, n, rmin, rmax, cmin, cmax, v0, out[0]);
        }
  return 0;
}
