// tile_common.h — the few definitions the two GEMM translation units share (gemm.hip: 8-wave / first-generation kernels and
// the dispatcher; lean.hip: the lean and wide kernel families).  Included INSIDE each unit's anonymous namespace.
#pragma once

constexpr int BK = 64;          // K elements per tile (128 bytes per row)
constexpr int ROW_BYTES = 128;



namespace g8 {
template <int N>
UDT_DEVINL void wait_vm() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// 16-byte LDS-DMA through a buffer descriptor: uniform base (SGPRs) + per-lane byte offset + scalar byte offset.
// A per-lane offset >= num_records (OOB) makes the load return zeros — used for rows past M / N and conv padding.
UDT_DEVINL void buf_lds16(__amdgpu_buffer_rsrc_t rsrc, void* lds_wave_base, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff,
                                           0, 0);
}
constexpr unsigned OOB = 0x80000000u;

}  // namespace g8

// hipFuncSetAttribute(max dynamic LDS) once per (kernel, device); thread-safe
struct AttrOnce {
  std::atomic<unsigned> done{0};
  hipError_t ensure(const void* fn, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = 0;
    if (done.load(std::memory_order_acquire) & (1u << dev)) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    // the persistent kernels assume >= 1 resident workgroup per CU with this much LDS; ask the runtime once
    done.fetch_or(1u << dev, std::memory_order_release);
    return hipSuccess;
  }
};

