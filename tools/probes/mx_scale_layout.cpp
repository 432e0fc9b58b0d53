// Probe: block-scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) A and B.
// C[m][n] = sum_k A[m][k] 2^(sa[m][k/32] - 127) * B[n][k] 2^(sb[n][k/32] - 127), one wave.
// Verified layout (this probe + mx_elem_block.cpp + mx_scale_diag.cpp, MI355X): the 64 k of an instruction are two scale blocks,
// k in [0, 32) and [32, 64).  Lane l = (row l % 32, half h = l / 32) holds, in registers 0..3, k = 16 h + [0, 16) (block 0) and,
// in registers 4..7, k = 32 + 16 h + [0, 16) (block 1) — fp8_mfma_layout.cpp's H1; the E8M0 byte `opsel` of the scale VGPR of
// the h = 0 lane of a row scales block 0 of that row, the h = 1 lane's byte scales block 1 — for A and for B alike.
// (With unit scales and the same k assignment on both operands, H0 = "32 consecutive k per lane" passes too: the sum over k
//  does not care.  With real block scales it does not: H0 gave garbage here.)
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/mx_scale_layout.cpp -o gpurun_out/mx_probe ; run on the GPU box
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

template <int SEL>
__global__ void probe(const uint8_t* A, const uint8_t* B, const uint32_t* SA, const uint32_t* SB, float* C) {
  const int lane = threadIdx.x;
  const int row = lane & 31, hi = lane >> 5;
  i32x8 a, b;
  for (int j = 0; j < 8; ++j) {
    const int k0 = (j < 4) ? (16 * hi + 4 * j) : (32 + 16 * hi + 4 * (j - 4));
    a[j] = *reinterpret_cast<const int*>(A + row * 64 + k0);
    b[j] = *reinterpret_cast<const int*>(B + row * 64 + k0);
  }
  const int sa = (int)SA[row * 2 + hi], sb = (int)SB[row * 2 + hi];
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, SEL, sa, SEL, sb);
  for (int r = 0; r < 16; ++r) {
    const int m = (r & 3) + 8 * (r >> 2) + 4 * hi;
    C[m * 32 + row] = c[r];
  }
}

int main() {
  uint8_t hA[32 * 64], hB[32 * 64];
  uint8_t ea[32 * 2], eb[32 * 2];
  srand(3);
  for (int i = 0; i < 32 * 64; ++i) {
    do { hA[i] = rand() & 0xff; } while ((hA[i] & 0x7f) == 0x7f || (hA[i] & 0x78) > 0x48);
    do { hB[i] = rand() & 0xff; } while ((hB[i] & 0x7f) == 0x7f || (hB[i] & 0x78) > 0x48);
  }
  for (int i = 0; i < 64; ++i) { ea[i] = 120 + rand() % 14; eb[i] = 122 + rand() % 10; }
  uint8_t *dA, *dB; uint32_t *dSA, *dSB; float* dC;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dSA, 64 * 4); hipMalloc(&dSB, 64 * 4); hipMalloc(&dC, 32 * 32 * 4);
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  double ref[32 * 32];
  double mag = 0;
  for (int m = 0; m < 32; ++m)
    for (int n = 0; n < 32; ++n) {
      double s = 0;
      for (int k = 0; k < 64; ++k)
        s += (double)e4m3_to_f(hA[m * 64 + k]) * ldexp(1.0, ea[m * 2 + k / 32] - 127) * (double)e4m3_to_f(hB[n * 64 + k]) *
             ldexp(1.0, eb[n * 2 + k / 32] - 127);
      ref[m * 32 + n] = s;
      mag = fmax(mag, fabs(s));
    }
  for (int sel = 0; sel < 4; ++sel) {
    uint32_t sa[64], sb[64];
    for (int i = 0; i < 64; ++i) {              // the selected byte carries the scale, the other three bytes are garbage
      sa[i] = 0x11223344u; sb[i] = 0x55667788u;
      sa[i] = (sa[i] & ~(0xffu << (8 * sel))) | ((uint32_t)ea[i] << (8 * sel));
      sb[i] = (sb[i] & ~(0xffu << (8 * sel))) | ((uint32_t)eb[i] << (8 * sel));
    }
    hipMemcpy(dSA, sa, sizeof(sa), hipMemcpyHostToDevice); hipMemcpy(dSB, sb, sizeof(sb), hipMemcpyHostToDevice);
    if (sel == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dC);
    if (sel == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dC);
    if (sel == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dC);
    if (sel == 3) hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dC);
    float out[32 * 32];
    hipMemcpy(out, dC, sizeof(out), hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < 32 * 32; ++i) err = fmax(err, fabs(out[i] - ref[i]));
    if (sel == 0) for (int i = 0; i < 6; ++i) printf("  out[%d] = %.6g  ref = %.6g  ratio %.4f\n", i * 37, out[i * 37], ref[i * 37], out[i * 37] / ref[i * 37]);
    printf("opsel %d: max|err| %.4g of max|ref| %.4g -> %s\n", sel, err, mag, err < 1e-3 * mag ? "MATCH" : "no");
  }
  return 0;
}
