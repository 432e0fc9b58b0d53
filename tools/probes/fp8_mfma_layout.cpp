// Probe: operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) A and B and unit scales.
// C[m][n] = sum_k A[m][k] * B[n][k], one wave.  Hypotheses for the 32 bytes (8 VGPRs) a lane holds:
//   H0: row = lane % 32, k = 32 * (lane / 32) + [0, 32)                      (32 consecutive k)
//   H1: row = lane % 32, k = {16 * (lane / 32) + [0, 16), 32 + 16 * (lane / 32) + [0, 16)}
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/fp8_mfma_layout.cpp -o gpurun_out/fp8_probe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

static float e4m3_to_f(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float r;
  if (e == 0) r = ldexpf((float)m / 8.f, -6);
  else if (e == 15 && m == 7) r = NAN;
  else r = ldexpf(1.f + (float)m / 8.f, e - 7);
  return s ? -r : r;
}

__global__ void probe(const uint8_t* A, const uint8_t* B, float* C, int hyp) {
  const int lane = threadIdx.x;
  const int row = lane & 31, hi = lane >> 5;
  i32x8 a, b;
  for (int j = 0; j < 8; ++j) {
    int k0;
    if (hyp == 0) k0 = 32 * hi + 4 * j;
    else k0 = (j < 4) ? (16 * hi + 4 * j) : (32 + 16 * hi + 4 * (j - 4));
    a[j] = *reinterpret_cast<const int*>(A + row * 64 + k0);
    b[j] = *reinterpret_cast<const int*>(B + row * 64 + k0);
  }
  f32x16 c = {0};
  // (a, b, c, cbsz = A format 0: fp8 e4m3, blgp = B format 0, opsel_a, scale_a (E8M0 127 = 1.0), opsel_b, scale_b)
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
  for (int r = 0; r < 16; ++r) {
    const int m = (r & 3) + 8 * (r >> 2) + 4 * hi;       // C/D layout of the 32x32 shapes: row from A, col = lane & 31
    C[m * 32 + row] = c[r];
  }
}

int main() {
  uint8_t hA[32 * 64], hB[32 * 64];
  srand(1);
  for (int i = 0; i < 32 * 64; ++i) {
    do { hA[i] = rand() & 0xff; } while ((hA[i] & 0x7f) == 0x7f || (hA[i] & 0x78) > 0x48);
    do { hB[i] = rand() & 0xff; } while ((hB[i] & 0x7f) == 0x7f || (hB[i] & 0x78) > 0x48);
  }
  float ref[32 * 32];
  for (int m = 0; m < 32; ++m)
    for (int n = 0; n < 32; ++n) {
      double s = 0;
      for (int k = 0; k < 64; ++k) s += (double)e4m3_to_f(hA[m * 64 + k]) * (double)e4m3_to_f(hB[n * 64 + k]);
      ref[m * 32 + n] = (float)s;
    }
  uint8_t *dA, *dB; float* dC;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(ref));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  for (int hyp = 0; hyp < 2; ++hyp) {
    float out[32 * 32];
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, hyp);
    hipMemcpy(out, dC, sizeof(out), hipMemcpyDeviceToHost);
    double err = 0, errT = 0, mag = 0;
    for (int m = 0; m < 32; ++m)
      for (int n = 0; n < 32; ++n) {
        err = fmax(err, fabs(out[m * 32 + n] - ref[m * 32 + n]));
        errT = fmax(errT, fabs(out[n * 32 + m] - ref[m * 32 + n]));
        mag = fmax(mag, fabs(ref[m * 32 + n]));
      }
    printf("hypothesis %d: max|err| %.4g (transposed C: %.4g) of max|ref| %.4g -> %s\n", hyp, err, errT, mag,
           err < 1e-3 * mag ? "MATCH" : errT < 1e-3 * mag ? "MATCH with C transposed" : "no");
  }
  return 0;
}
