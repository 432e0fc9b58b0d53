// Do matrix (MFMA) and vector (VALU) instructions of DIFFERENT waves on the same SIMD execute concurrently on gfx950?  And MFMAs with
// independent VALU instructions of the SAME wave?  One workgroup of 8 waves on one CU: waves w and w + 4 share SIMD w.  Modes:
//   0: waves 0..3 run a chain-free MFMA loop (4 rotating accumulators), waves 4..7 exit           -> T_mfma
//   1: waves 4..7 run a VALU loop (8 independent v_fma_f32 chains), waves 0..3 exit               -> T_valu
//   2: both                                                                                        -> max (overlap) or sum (no overlap)?
//   3: waves 0..3 run BOTH loops interleaved in one instruction stream (1 MFMA : R VALU), waves 4..7 exit
// for the bf16 32x32x16 MFMA (8 passes) and the f8f6f4 32x32x64 one (16 passes).  Prints s_memtime ticks per wave.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_valu_overlap.cpp -o tools/probes/bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
constexpr int ITERS = 2048;
template <bool F8>
__device__ void mfma_loop(f32x16 (&acc)[4], float seed) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
  i32x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = 0x38383838; b8[i] = 0x38383838; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (F8) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[j], 0, 0, 0, 127, 0, 127);
      else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
  }
}
__device__ void valu_loop(float (&r)[8], int n) {
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[j]));
  }
}
template <bool F8, int R>
__device__ void both_loop(f32x16 (&acc)[4], float (&r)[8]) {
  i32x8 a8, b8;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a8[i] = 0x38383838; b8[i] = 0x38383838; a[i] = (__bf16)1.0f; b[i] = (__bf16)1.0f; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (F8) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[j], 0, 0, 0, 127, 0, 127);
      else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < R; ++k) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[k & 7]));
    }
  }
}
template <bool F8>
__global__ void __launch_bounds__(512) probe(int mode, int valu_iters, float* out, long long* ticks) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  float r[8];
  for (int j = 0; j < 8; ++j) r[j] = 1e-3f * threadIdx.x + j;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    if (mode == 0 || mode == 2) mfma_loop<F8>(acc, 1.0f);
    if (mode == 3) both_loop<F8, F8 ? 12 : 6>(acc, r);
  } else {
    if (mode == 1 || mode == 2) valu_loop(r, valu_iters);
  }
  asm volatile("s_nop 0" ::: "memory");
  float sink = 0.f;
  for (int j = 0; j < 4; ++j) sink += acc[j][0] + acc[j][7];
  for (int j = 0; j < 8; ++j) sink += r[j];
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = sink;
  if ((threadIdx.x & 63) == 0) ticks[wave] = t1 - t0;
}
int main() {
  float* o; long long* t;
  hipMalloc(&o, 512 * 4); hipMalloc(&t, 8 * 8);
  for (int f8 = 0; f8 < 2; ++f8) {
    // VALU loop sized to take about as long as the MFMA loop: ITERS * 4 MFMAs * (8 | 16) passes * 4 cycles = that many cycles; 8 v_fma = 32 cycles
    const int valu_iters = ITERS * 4 * (f8 ? 16 : 8) * 4 / 32;
    printf("%s: %d MFMAs per wave; VALU loop of %d x 8 v_fma_f32\n", f8 ? "v_mfma_scale_f32_32x32x64_f8f6f4 (16 passes)" : "v_mfma_f32_32x32x16_bf16 (8 passes)", ITERS * 4, valu_iters);
    for (int mode = 0; mode < 4; ++mode) {
      long long h[8];
      for (int rep = 0; rep < 2; ++rep) {
        if (f8) hipLaunchKernelGGL(probe<true>, dim3(1), dim3(512), 0, 0, mode, valu_iters, o, t);
        else hipLaunchKernelGGL(probe<false>, dim3(1), dim3(512), 0, 0, mode, valu_iters, o, t);
        hipDeviceSynchronize();
      }
      hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
      printf("  mode %d: wave ticks", mode);
      for (int w = 0; w < 8; ++w) printf(" %8lld", h[w]);
      printf("\n");
    }
  }
  return 0;
}
