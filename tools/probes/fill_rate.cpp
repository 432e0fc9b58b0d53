// Probe: what rate can a CU fill its LDS at through 16-byte LDS-DMA (buffer_load ... lds), as a function of
//   waves per workgroup (4 / 8), workgroups per CU (LDS footprint), K-tiles in flight, and where the data comes from
//   (every workgroup re-reads the same 2 MiB -> L2-resident; or each workgroup streams its own rows -> HBM).
// Each workgroup "stages" ITERS K-tiles of kROWS rows x 128 B (like a GEMM's A+B tile) into a ring of kNST stages with the
// lean kernels' loop (counted vmcnt, one barrier per tile), no MFMAs.  Prints GB/s per CU and in aggregate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
namespace {
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int kNW, int kROWS, int kNST>
__device__ __forceinline__ void fill_body(const char* src, long long wg_stride, int row_bytes, int iters, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PIECES = kROWS / 8, PW = PIECES / kNW, STAGE = kROWS * 128;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const char* base = src + (long long)blockIdx.x * wg_stride;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0x7fffffff, 0x00020000);
  unsigned voff[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int row = (wave + kNW * i) * 8 + (lane >> 3);
    voff[i] = (unsigned)((long long)row * row_bytes + (lane & 7) * 16);
  }
  auto stage = [&](int st, int kt) {
#pragma unroll
    for (int i = 0; i < PW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + st * STAGE + (wave + kNW * i) * 1024), 16, voff[i], kt * 128, 0, 0);
  };
#pragma unroll
  for (int s = 0; s < kNST - 1; ++s) stage(s, s);
  int st = 0, acc = 0;
  for (int kt = 0; kt < iters; ++kt) {
    if (kNST > 2 && kt + kNST - 2 < iters) wait_vm<PW * (kNST > 2 ? kNST - 2 : 0)>(); else wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + kNST - 1 < iters) { int s2 = st + kNST - 1; if (s2 >= kNST) s2 -= kNST; stage(s2, kt + kNST - 1); }
    acc += *reinterpret_cast<const int*>(smem + st * STAGE + tid * 16);     // one read per tile keeps the data "used"
    st = st + 1 == kNST ? 0 : st + 1;
  }
  if (acc == 0x12345678) sink[0] = acc;
}
__global__ void __launch_bounds__(256) fill_4_256_2(const char* src, long long wg_stride, int row_bytes, int iters, int* sink) { fill_body<4, 256, 2>(src, wg_stride, row_bytes, iters, sink); }
__global__ void __launch_bounds__(256) fill_4_256_3(const char* src, long long wg_stride, int row_bytes, int iters, int* sink) { fill_body<4, 256, 3>(src, wg_stride, row_bytes, iters, sink); }
__global__ void __launch_bounds__(256) fill_4_256_4(const char* src, long long wg_stride, int row_bytes, int iters, int* sink) { fill_body<4, 256, 4>(src, wg_stride, row_bytes, iters, sink); }
__global__ void __launch_bounds__(512) fill_8_384_3(const char* src, long long wg_stride, int row_bytes, int iters, int* sink) { fill_body<8, 384, 3>(src, wg_stride, row_bytes, iters, sink); }
__global__ void __launch_bounds__(512) fill_8_512_2(const char* src, long long wg_stride, int row_bytes, int iters, int* sink) { fill_body<8, 512, 2>(src, wg_stride, row_bytes, iters, sink); }
__global__ void __launch_bounds__(256) fill_4_128_2(const char* src, long long wg_stride, int row_bytes, int iters, int* sink) { fill_body<4, 128, 2>(src, wg_stride, row_bytes, iters, sink); }
__global__ void __launch_bounds__(256) fill_4_128_4(const char* src, long long wg_stride, int row_bytes, int iters, int* sink) { fill_body<4, 128, 4>(src, wg_stride, row_bytes, iters, sink); }
__global__ void __launch_bounds__(512) fill_8_256_4(const char* src, long long wg_stride, int row_bytes, int iters, int* sink) { fill_body<8, 256, 4>(src, wg_stride, row_bytes, iters, sink); }
}  // namespace
struct Res { float ms; };
#define RUN(NWV, ROWSV, NSTV, NAME, WGS, L2V)                                                                                   \
  do {                                                                                                                        \
    const int cus = 256, G = cus * (WGS), iters = 64, row_bytes = iters * 128;                                                 \
    const int smem = (NSTV) * (ROWSV) * 128;                                                                                  \
    auto kern = fill_##NWV##_##ROWSV##_##NSTV;                                                                                 \
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);               \
    const long long per_wg = (long long)(ROWSV) * row_bytes;                                                                  \
    const long long stride = (L2V) ? 0 : per_wg;                                                                              \
    if (!(L2V) && (size_t)(per_wg * G) > bytes) { printf("%s: buffer too small\n", NAME); break; }                            \
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);                                                              \
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(G), dim3((NWV) * 64), smem, 0, buf, stride, row_bytes, iters, sink); \
    hipEventRecord(e0);                                                                                                       \
    const int reps = 10;                                                                                                      \
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(G), dim3((NWV) * 64), smem, 0, buf, stride, row_bytes, iters, sink); \
    hipEventRecord(e1); hipEventSynchronize(e1);                                                                              \
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;                                                                   \
    const double total = (double)per_wg * G;                                                                                  \
    printf("%-44s %s wg/cu=%d lds=%3dK: %8.1f us  %7.1f GB/s per CU  %6.2f TB/s total\n", NAME, (L2V) ? "L2 " : "HBM", WGS,     \
           smem / 1024, ms * 1e3, total / cus / ms / 1e6, total / ms / 1e9);                                                   \
  } while (0)
int main() {
  size_t bytes = (size_t)3 << 30;
  char* buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
  int* sink; hipMalloc(&sink, 64);
  for (int l2 = 1; l2 >= 0; --l2) {
    RUN(4, 256, 2, "4 waves, 256 rows (32 KiB tile), 2 stages", 1, l2);
    RUN(4, 256, 2, "4 waves, 256 rows (32 KiB tile), 2 stages", 2, l2);
    RUN(4, 256, 3, "4 waves, 256 rows (32 KiB tile), 3 stages", 1, l2);
    RUN(4, 256, 4, "4 waves, 256 rows (32 KiB tile), 4 stages", 1, l2);
    RUN(8, 384, 3, "8 waves, 384 rows (48 KiB tile), 3 stages", 1, l2);
    RUN(8, 512, 2, "8 waves, 512 rows (64 KiB tile), 2 stages", 1, l2);
    RUN(4, 128, 2, "4 waves, 128 rows (16 KiB tile), 2 stages", 2, l2);
    RUN(4, 128, 2, "4 waves, 128 rows (16 KiB tile), 2 stages", 4, l2);
    RUN(4, 128, 4, "4 waves, 128 rows (16 KiB tile), 4 stages", 2, l2);
    RUN(8, 256, 4, "8 waves, 256 rows (32 KiB tile), 4 stages", 1, l2);
  }
  return 0;
}
