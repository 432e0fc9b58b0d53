// common.h — device helpers shared by the gfx950 kernels (wave64, MFMA, LDS-DMA staging).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/udt_kernels.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // one MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define UDT_DEVINL __device__ __forceinline__

// cost-attribution switches of the GEMM finishing code (udt_debug_set "no_*"): they produce WRONG results and exist
// only in measurement builds (hipcc -DUDT_MEASURE); in the product library every test folds to 0
#ifdef UDT_MEASURE
#define UDT_DBG(flags, bit) ((((flags) >> (bit)) & 1) != 0)
#else
#define UDT_DBG(flags, bit) false
#endif

UDT_DEVINL float bf16_bits_to_f32(uint32_t v) { return __uint_as_float(v << 16); }
UDT_DEVINL float bf16_lo(uint32_t packed) { return __uint_as_float(packed << 16); }
UDT_DEVINL float bf16_hi(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }

// round-to-nearest-even fp32 -> bf16 pair packed in one dword (lo = first element)
UDT_DEVINL uint32_t pack_bf16x2(float lo, float hi) {
  f32x2 v = {lo, hi};
  bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
  return __builtin_bit_cast(uint32_t, b);
}

UDT_DEVINL float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
// x * sigmoid(x) on v_exp_f32 / v_rcp_f32 (1 ulp): the IEEE division sequence costs ~10 VALU ops per element
UDT_DEVINL float silu_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-x * 1.4426950408889634f));
}
// exact-erf GELU (F.gelu default).  erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below bf16
// resolution) on v_rcp_f32 / v_exp_f32: ~15 VALU ops instead of libm erff's branchy ~40.
UDT_DEVINL float gelu_erf_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  float poly = 1.061405429f;
  poly = poly * t - 1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t - 0.284496736f;
  poly = poly * t + 0.254829592f;
  poly = poly * t;
  const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
  const float erf_abs = 1.0f - poly * e;
  const float erf = __builtin_copysignf(erf_abs, x);
  return 0.5f * x * (1.0f + erf);
}

// GEGLU output x * GELU(g) from PRE-SCALED operands xs = GEGLU_XS * x, gs = GEGLU_GS * g (the callers fold the two constants into
// their bias / LayerNorm / alpha multipliers), exact-erf GELU (F.gelu default):
//   x GELU(g) = 0.5 x g + 0.5 x |g| erf(|g| / sqrt 2) = xs gs + xs |gs| (1 - erfc(z)),   z = |g| / sqrt 2 = |gs| / sqrt(log2 e)
// with erfc(z) = 2^P(|gs|): log2 erfc is smooth and nearly quadratic, a degree-6 polynomial through the origin (least squares on
// Chebyshev nodes of z in [0, 6], weighted by erfc, so that the ABSOLUTE error of erfc is what is minimised) is within 2.9e-7 of
// erfc — the accuracy of Abramowitz-Stegun 7.1.26 (1.5e-7), which gelu_erf_f uses, without its reciprocal: 11 VALU instructions, ONE
// transcendental, against 13 with two for the A-S form on the same pre-scaled operands (and 17 + 2 behind un-scaled ones).  fp32
// evaluation over g in [-12, 12]: max |error| / (|x g| + 1e-3) = 3.5e-7, identical to the A-S form's 3.4e-7 (tools/fit_erfc_poly.py
// regenerates the coefficients and the comparison).  |gs| is clamped at z = 6 (erfc < 2e-17) before the polynomial.
// The hot GEGLU epilogues (lean.h, rowres.h) are bound by this arithmetic (profiles/r04_rowres.txt, r04_rowmlp_rejected.txt).
constexpr float GEGLU_GS = 0.70710678118654752f * 1.2011224087864498f;     // 1/sqrt(2) * sqrt(log2 e)
constexpr float GEGLU_XS = 0.5f / GEGLU_GS;
UDT_DEVINL float geglu_scaled(float xs, float gs) {
  const float t = fminf(fabsf(gs), 7.2067347f);
  float P = 5.1346913e-05f;
  P = P * t - 0.001489955f;
  P = P * t + 0.014895574f;
  P = P * t - 0.086436f;
  P = P * t - 0.636406f;
  P = P * t - 1.3553386f;
  P = P * t;
  const float e = __builtin_amdgcn_exp2f(P);              // erfc(|g| / sqrt 2)
  const float ha = xs * fabsf(gs);
  const float hs = __builtin_fmaf(xs, gs, ha);
  return __builtin_fmaf(-ha, e, hs);
}

// 16-byte global -> LDS DMA.  `lds_wave_base` must be wave-uniform; lane l lands at base + 16*l.
UDT_DEVINL void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// 16-byte write-through (sc1) store: the data is visible at agent scope (other XCDs' L2s) once the wave's vmcnt
// drains, so a flag can be raised after it WITHOUT a release fence — the fence's buffer_wbl2 writes back the whole
// L2 and was measured at 4-5 us per episode (cdna_hip_programming.md, split-K recipe, "sc1 slab stores").
UDT_DEVINL void store16_sc1(void* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

UDT_DEVINL void wait_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

UDT_DEVINL bf16x8_t lds_read_frag(const char* p) { return *reinterpret_cast<const bf16x8_t*>(p); }

UDT_DEVINL f32x16 mfma32(bf16x8_t a, bf16x8_t b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// fp8 (OCP e4m3) operands: 32 bytes per lane = 64 K-elements per instruction.  The only large-K fp8 MFMA on gfx950 is the
// block-scaled form (cdna_hip_programming.md §3).
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
UDT_DEVINL i32x8_t lds_read_frag32(const char* p0, const char* p1) {
  const u32x4 a = *reinterpret_cast<const u32x4*>(p0);
  const u32x4 b = *reinterpret_cast<const u32x4*>(p1);
  i32x8_t r = {(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
  return r;
}

// MX (microscaling) form of the same instruction.  The 64 k of one instruction are TWO scale blocks, k in [0, 32) and [32, 64);
// lane (row l31, half h) holds k = 16 h + [0, 16) in registers 0..3 and k = 32 + 16 h + [0, 16) in registers 4..7, and byte OPSEL of
// the scale VGPR of the row's h = 0 lane scales block 0, the h = 1 lane's byte block 1 (tools/probes/mx_scale_layout.cpp,
// mx_elem_block.cpp, mx_scale_diag.cpp — measured on MI355X; NOT "a lane scales its own 32 elements").  Here the ACTIVATION
// fragment (the MFMA's B operand, see gemm.hip: the weight fragment is the A operand) carries the block scales, the weight side
// keeps the unit scale (its per-output-channel fp32 scale is applied to the accumulators).  OPSEL is an instruction immediate.
template <int OPSEL>
UDT_DEVINL f32x16 mfma32_mx8(i32x8_t w, i32x8_t x, f32x16 c, int scale) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w, x, c, 0, 0, 0, 127, OPSEL, scale);
}

// ---- MX8 activations: OCP e4m3 elements + one E8M0 scale per 32 consecutive K-elements of a row ---------------------------
// Layout in memory (udt_gemm_desc.a_scale / q8_scale): elements [M][K] bytes; scales as uint32 [K / 128][M] — dword (kt, m)
// holds the four block scales of K-tile kt of row m (byte j = block 4 kt + j), so a consumer lane fetches the scales of a whole
// 128-element LDS row with one aligned, row-coalesced dword load.
// The scale of a block is the smallest power of two 2^(s-127) with amax / 2^(s-127) <= 448 (the largest e4m3): the conversion
// never saturates, and every block uses the top two binades of the format.
UDT_DEVINL uint32_t mx8_scale_byte(float amax) {
  // amax <= 1.75 * 2^p  <=>  amax * 4/7 <= 2^p; the constant sits a few ulps ABOVE 4/7, so a boundary case picks the coarser scale
  const uint32_t e = (__float_as_uint(amax * 0.57142866f) >> 23) & 0xffu;
  return e > 7u ? e - 7u : 0u;
}
UDT_DEVINL float mx8_inv_scale(uint32_t s) { return __uint_as_float((254u - s) << 23); }     // 2^(127 - s), s <= 247
UDT_DEVINL uint32_t mx8_pack4(float a, float b, float c, float d) {
  int r = 0;
  r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, r, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (uint32_t)r;
}
template <int CTRL>
UDT_DEVINL float dpp_max(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false);
  return fmaxf(v, __builtin_bit_cast(float, t));
}
// Row layout (a lane owns 8 consecutive columns, the 4 lanes of an aligned quad own one 32-column block): quantise the block;
// returns the lane's 8 bytes, `sbyte` = the block's scale (the same in all four lanes)
UDT_DEVINL u32x2 mx8_quant_row8(const float (&o)[8], uint32_t& sbyte) {
  float a = fmaxf(fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))),
                  fmaxf(fmaxf(fabsf(o[4]), fabsf(o[5])), fmaxf(fabsf(o[6]), fabsf(o[7]))));
  a = dpp_max<0xB1>(a);       // lane ^ 1
  a = dpp_max<0x4E>(a);       // lane ^ 2
  sbyte = mx8_scale_byte(a);
  const float m = mx8_inv_scale(sbyte);
  u32x2 r = {mx8_pack4(o[0] * m, o[1] * m, o[2] * m, o[3] * m), mx8_pack4(o[4] * m, o[5] * m, o[6] * m, o[7] * m)};
  return r;
}
// The same 8 values with a FIXED multiplier (no block scale: unit E8M0), clamped to the e4m3 range — the conversion does not saturate
UDT_DEVINL u32x2 e4m3_fixed_row8(const float (&o)[8], float m) {
  float c[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) c[j] = fminf(fmaxf(o[j] * m, -448.f), 448.f);
  u32x2 r = {mx8_pack4(c[0], c[1], c[2], c[3]), mx8_pack4(c[4], c[5], c[6], c[7])};
  return r;
}
// Accumulator layout of a 32 x 32 MFMA tile whose D rows are the CHANNELS (lane = (row l31, half hi) holds channels
// 8 q + 4 hi + e, q = 0..3, e = 0..3, of its row's 32-channel block; the other 16 sit in lane ^ 32): quantise the block;
// out[q] = the 4 bytes of channels 8 q + 4 hi .. + 3
UDT_DEVINL void mx8_quant_acc16(const float (&v)[16], uint32_t (&out)[4], uint32_t& sbyte) {
  float a = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) a = fmaxf(a, fabsf(v[r]));
  const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, a), false, false);
  a = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
  sbyte = mx8_scale_byte(a);
  const float m = mx8_inv_scale(sbyte);
#pragma unroll
  for (int q = 0; q < 4; ++q) out[q] = mx8_pack4(v[q * 4] * m, v[q * 4 + 1] * m, v[q * 4 + 2] * m, v[q * 4 + 3] * m);
}

// workgroup -> iteration-range index.  Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8,
// speed heuristic only); give every XCD a contiguous slice of the iteration space so that its private L2 sees
// neighbouring tiles (which share the weight column tile).
UDT_DEVINL int range_index(int g, int G) {
  if ((G & 7) != 0) return g;
  return (g & 7) * (G >> 3) + (g >> 3);
}

// ---- transposed LDS reads (ds_read_b64_tr_b16 / _b8) through inline assembly -----------------------------------------------
// hipcc's waitcnt insertion treats the __builtin_amdgcn_ds_read_tr* intrinsics as LDS reads that may alias ANY LDS-DMA load in
// flight and puts `s_waitcnt vmcnt(0)` in front of the first one: in a ring of DMA-staged tiles that drains the loads issued for the
// tile two steps ahead in the middle of every step (found in all three flash attention kernels, round 5).  An asm statement is
// opaque to that pass: the kernel's own vmcnt(N) + barrier at the top of a step order the DMA against these reads.  The result
// registers must pass through lds_tr_wait() before use (the pass does not count the asm's LDS operation either).
UDT_DEVINL unsigned lds_offset(const void* p) { return (unsigned)(uintptr_t)p; }       // (the low half of a flat LDS address)
// (off: a value that is a compile-time constant once the surrounding loops are unrolled — it becomes the instruction's offset field)
UDT_DEVINL u32x2 lds_tr16_b64(unsigned addr, int off) {
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(off) : "memory");
  return r;
}
UDT_DEVINL u32x2 lds_tr8_b64(unsigned addr, int off) {
  u32x2 r;
  asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(off) : "memory");
  return r;
}
UDT_DEVINL void lds_tr_wait(u32x2& a, u32x2& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) : : "memory"); }
UDT_DEVINL void lds_tr_wait(u32x2& a, u32x2& b, u32x2& c, u32x2& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory");
}
// max of a value with its partner lane ^ 32 in the VALU (v_permlane32_swap; __shfl_xor(v, 32) is a ds_bpermute round trip)
UDT_DEVINL float half_swap_max(float v) {
  const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
}
UDT_DEVINL float half_swap_sum(float v) {
  const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  return __builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]);
}

// ---- cross-lane sums without LDS traffic ---------------------------------------------------------------------------
// DPP row operations (quad_perm / row_ror inside a row of 16 lanes) and the gfx950 row / half swaps add a value over
// lane groups in the VALU — no ds_bpermute round trips (measured: a __shfl_xor butterfly over the 32 row lanes of an
// accumulator tile cost ~7 us per launch in the convolution epilogues).
template <int CTRL>
UDT_DEVINL float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false);
  return v + __builtin_bit_cast(float, t);
}
// sum over the 16 lanes of each DPP row (lanes 16k .. 16k+15), result in every lane of the row
UDT_DEVINL float row16_sum(float v) {
  v = dpp_add<0xB1>(v);       // quad_perm [1,0,3,2]: lane ^ 1
  v = dpp_add<0x4E>(v);       // quad_perm [2,3,0,1]: lane ^ 2
  v = dpp_add<0x124>(v);      // row_ror:4
  v = dpp_add<0x128>(v);      // row_ror:8
  return v;
}
// + the neighbouring row (lane ^ 16): v_permlane16_swap exchanges the odd rows of its first operand with the even
// rows of the second; with both operands = v the two results are [r0 r0 r2 r2] and [r1 r1 r3 r3]
UDT_DEVINL float xor16_sum(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
// + the other half of the wave (lane ^ 32)
UDT_DEVINL float xor32_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

// ---- host-side helpers -------------------------------------------------------------------------
int udt_set_hip_error(hipError_t e);   // records e, returns UDT_ERR_HIP (or UDT_OK when e == success)
const uint16_t* udt_zero_page();       // >= 4 KiB of zeroed memory on the current device (lazily allocated once per device)

struct UdtProfScope {                  // brackets a launch with events when profiling is enabled
  int cls; hipStream_t s; void* rec;
  UdtProfScope(int cls_, hipStream_t s_);
  ~UdtProfScope();
};

void udt_prof_tag(void* rec, const char* tag);   // attach a shape description to a profiled launch

#define UDT_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t _e = hipGetLastError();                       \
    if (_e != hipSuccess) return udt_set_hip_error(_e);      \
  } while (0)
