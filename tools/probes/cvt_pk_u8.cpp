// Probe of v_cvt_pk_u8_f32 (__builtin_amdgcn_cvt_pk_u8_f32) and of VALU issue rates on gfx950, for an e4m3 softmax that writes the
// probabilities' e4m3 BYTES directly (an e4m3 bit pattern is a piecewise-linear log2 scale: byte = 8 (log2 p + 7) + mantissa
// correction): (1) rounding / saturation / NaN behaviour of the float -> u8 conversion, byte select semantics;
// (2) cycles per wave64 instruction of v_exp_f32, v_cvt_pk_u8_f32, v_pk_fma_f32, v_max3_f32, v_cvt_pk_fp8_f32 — dependent-free chains
// of 8 independent registers, 4096 instructions, one wave per SIMD (s_memtime around the loop).
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/cvt_pk_u8.cpp -o tools/probes/bin/cvt_pk_u8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void sem(const float* in, unsigned* out, int n) {
  const int i = threadIdx.x;
  if (i < n) {
    out[i * 2] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0u, 0xAABBCCDDu);
    out[i * 2 + 1] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 2u, 0xAABBCCDDu);
  }
}
template <int OP>
__global__ void rate(float* out, long long* cyc, float seed) {
  float r[8];
  for (int j = 0; j < 8; ++j) r[j] = seed + threadIdx.x * 1e-3f + j;
  unsigned u[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  f2 p[8];
  for (int j = 0; j < 8; ++j) p[j] = f2{r[j], r[j] + 1.f};
  const long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int it = 0; it < 512; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(r[j]));
      if (OP == 1) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u[j]) : "v"(r[j]));
      if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[j]));
      if (OP == 3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[j]) : "v"(r[(j + 1) & 7]), "v"(r[(j + 2) & 7]));
      if (OP == 4) asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(u[j]) : "v"(r[j]), "v"(r[(j + 1) & 7]));
      if (OP == 5) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[j]));
      if (OP == 6) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(u[j]) : "v"(r[j]));
    }
  }
  asm volatile("s_nop 0" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0.f;
  for (int j = 0; j < 8; ++j) acc += r[j] + (float)u[j] + p[j][0] + p[j][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  const float vals[] = {0.f, 0.49f, 0.5f, 0.51f, 1.49f, 1.5f, 2.5f, 3.5f, 119.99f, 120.5f, 254.5f, 255.4f, 255.6f, 300.f, 1e9f, -0.4f, -0.6f, -1.f, -1e9f,
                        INFINITY, -INFINITY, NAN};
  const int n = sizeof(vals) / sizeof(float);
  float* din; unsigned* dout;
  hipMalloc(&din, sizeof(vals)); hipMalloc(&dout, n * 8);
  hipMemcpy(din, vals, sizeof(vals), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, din, dout, n);
  unsigned h[64];
  hipMemcpy(h, dout, n * 8, hipMemcpyDeviceToHost);
  printf("v_cvt_pk_u8_f32(x, sel, 0xAABBCCDD)\n");
  for (int i = 0; i < n; ++i) printf("  x = %12g  sel 0 -> %08x (byte %3u)   sel 2 -> %08x\n", vals[i], h[i * 2], h[i * 2] & 0xff, h[i * 2 + 1]);
  float* o; long long* c;
  hipMalloc(&o, 64 * 4 * 4); hipMalloc(&c, 8 * 8);
  const char* names[] = {"v_exp_f32", "v_cvt_pk_u8_f32", "v_pk_fma_f32", "v_max3_f32", "v_cvt_pk_fp8_f32", "v_fma_f32", "v_cvt_u32_f32"};
  for (int op = 0; op < 7; ++op) {
    for (int rep = 0; rep < 2; ++rep) {
      switch (op) {
        case 0: hipLaunchKernelGGL(rate<0>, dim3(1), dim3(64), 0, 0, o, c, 0.001f); break;
        case 1: hipLaunchKernelGGL(rate<1>, dim3(1), dim3(64), 0, 0, o, c, 0.001f); break;
        case 2: hipLaunchKernelGGL(rate<2>, dim3(1), dim3(64), 0, 0, o, c, 0.001f); break;
        case 3: hipLaunchKernelGGL(rate<3>, dim3(1), dim3(64), 0, 0, o, c, 0.001f); break;
        case 4: hipLaunchKernelGGL(rate<4>, dim3(1), dim3(64), 0, 0, o, c, 0.001f); break;
        case 5: hipLaunchKernelGGL(rate<5>, dim3(1), dim3(64), 0, 0, o, c, 0.001f); break;
        case 6: hipLaunchKernelGGL(rate<6>, dim3(1), dim3(64), 0, 0, o, c, 0.001f); break;
      }
      hipDeviceSynchronize();
    }
    long long hc; hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    // s_memtime counts at the constant 100 MHz reference clock on gfx9: report raw ticks per instruction and the ratio to v_fma_f32 below
    printf("%-18s %8lld ticks for 4096 wave64 instructions (%.4f ticks each)\n", names[op], hc, hc / 4096.0);
  }
  return 0;
}
