// lean.hip — the lean (co-resident, lean.h) and wide (wide.h) MFMA kernel families and their launchers; planned and
// dispatched by gemm.hip (lean_plan / lean_conv_plan), which hands over filled argument blocks (lean_params.h).
// A translation unit of its own so that the two halves of the GEMM code compile in parallel.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <atomic>

namespace {
#include "tile_common.h"
#include "lean_params.h"
#include "lean.h"
#include "wide.h"
#include "rowres.h"

template <int NW, int WGM, int WGN, int TM, int TN, int NST, int TMB = TM>
hipError_t launch_lean(const lg::LParams& lp, int smem, int G, bool geglu, bool ln, hipStream_t s) {
  static AttrOnce once[5];
  const void* fn;
  const bool stats = lp.colstats != nullptr;
  if constexpr (NW == 4 && NST == 2) {                    // (the two kernels with a statistics-emitting epilogue)
    if (stats) {
      fn = (const void*)lg::lgemm_kernel<NW, WGM, WGN, TM, TN, NST, false, false, TMB, true>;
      hipError_t e = once[4].ensure(fn, smem);
      if (e != hipSuccess) return e;
      void* args[] = {const_cast<lg::LParams*>(&lp)};
      return hipLaunchKernel(fn, dim3(G), dim3(NW * 64), args, smem, s);
    }
  }
  if (stats) return hipErrorInvalidValue;
  if constexpr (TN == 2) {
    fn = geglu ? (ln ? (const void*)lg::lgemm_kernel<NW, WGM, WGN, TM, TN, NST, true, true, TMB> : (const void*)lg::lgemm_kernel<NW, WGM, WGN, TM, TN, NST, true, false, TMB>)
               : (ln ? (const void*)lg::lgemm_kernel<NW, WGM, WGN, TM, TN, NST, false, true, TMB> : (const void*)lg::lgemm_kernel<NW, WGM, WGN, TM, TN, NST, false, false, TMB>);
  } else {
    fn = ln ? (const void*)lg::lgemm_kernel<NW, WGM, WGN, TM, TN, NST, false, true, TMB> : (const void*)lg::lgemm_kernel<NW, WGM, WGN, TM, TN, NST, false, false, TMB>;
  }
  hipError_t e = once[(geglu ? 2 : 0) + (ln ? 1 : 0)].ensure(fn, smem);
  if (e != hipSuccess) return e;
  void* args[] = {const_cast<lg::LParams*>(&lp)};
  return hipLaunchKernel(fn, dim3(G), dim3(NW * 64), args, smem, s);
}

// MX8 instances (lean.h FP8 / EMIT; BASELINE config #5): the 128 x 128 / two-workgroups-per-CU configuration.
//   emit, bf16 operands : plain epilogue + the result again as an MX8 activation (proj_in -> q|k|v)
//   fp8 operands        : plain (to_out / proj_out, optionally with column statistics), plain + emit (ff.net[2] -> proj_out),
//                         LayerNorm-folded plain (q|k|v), LayerNorm-folded GEGLU + emit (ff.net[0] -> ff.net[2])
template <bool GEGLU, bool LN, bool STATS, bool FP8, bool EMIT>
hipError_t launch_lean_mx8_k(const lg::LParams& lp, int smem, int G, hipStream_t s) {
  static AttrOnce once;
  const void* fn = (const void*)lg::lgemm_kernel<4, 2, 2, 2, 2, 2, GEGLU, LN, 2, STATS, FP8, EMIT>;
  hipError_t e = once.ensure(fn, smem);
  if (e != hipSuccess) return e;
  void* args[] = {const_cast<lg::LParams*>(&lp)};
  return hipLaunchKernel(fn, dim3(G), dim3(256), args, smem, s);
}
hipError_t launch_lean_mx8(const lg::LParams& lp, int smem, int G, bool geglu, bool ln, bool fp8, bool emit, hipStream_t s) {
  const bool stats = lp.colstats != nullptr;
  if (!fp8) {
    if (emit && !geglu && !ln && !stats) return launch_lean_mx8_k<false, false, false, false, true>(lp, smem, G, s);
    return hipErrorInvalidValue;
  }
  if (geglu) return (ln && emit && !stats) ? launch_lean_mx8_k<true, true, false, true, true>(lp, smem, G, s) : hipErrorInvalidValue;
  if (ln) return stats ? hipErrorInvalidValue
                 : emit ? launch_lean_mx8_k<false, true, false, true, true>(lp, smem, G, s)      // q|k|v whose consumer is the e4m3 attention
                        : launch_lean_mx8_k<false, true, false, true, false>(lp, smem, G, s);
  if (stats) return emit ? hipErrorInvalidValue : launch_lean_mx8_k<false, false, true, true, false>(lp, smem, G, s);
  return emit ? launch_lean_mx8_k<false, false, false, true, true>(lp, smem, G, s) : launch_lean_mx8_k<false, false, false, true, false>(lp, smem, G, s);
}

template <int TW, int TH, bool UPS, bool STATS>
hipError_t launch_lconv3s(const lg::C3Params& c3, hipStream_t s) {
  static AttrOnce once;
  constexpr int smem = lg::C3Geo<TW, TH, UPS>::SMEM;
  hipError_t e = once.ensure(reinterpret_cast<const void*>(lg::lconv3_kernel<TW, TH, UPS, STATS>), smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((lg::lconv3_kernel<TW, TH, UPS, STATS>), dim3(c3.G), dim3(256), smem, s, c3);
  return hipGetLastError();
}
template <int TW, int TH, bool UPS>
hipError_t launch_lconv3(const lg::C3Params& c3, hipStream_t s) {
  return c3.colstats ? launch_lconv3s<TW, TH, UPS, true>(c3, s) : launch_lconv3s<TW, TH, UPS, false>(c3, s);
}

template <bool STATS>
hipError_t launch_wconv3s(const lg::C3Params& c3, hipStream_t s) {
  static AttrOnce once;
  constexpr int smem = wd::WGeo::SMEM;
  hipError_t e = once.ensure(reinterpret_cast<const void*>(wd::wconv3_kernel<STATS>), smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((wd::wconv3_kernel<STATS>), dim3(c3.G), dim3(256), smem, s, c3);
  return hipGetLastError();
}
template <int KT, bool GEGLU, bool EMIT = false>
hipError_t launch_rowres(const lg::LParams& lp, int G, hipStream_t s) {
  static AttrOnce once;
  constexpr int smem = rr::RGeo<KT, GEGLU>::SMEM;
  hipError_t e = once.ensure(reinterpret_cast<const void*>(rr::rgemm_kernel<KT, GEGLU, EMIT>), smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((rr::rgemm_kernel<KT, GEGLU, EMIT>), dim3(G), dim3(256), smem, s, lp);
  return hipGetLastError();
}

hipError_t launch_wconv3(const lg::C3Params& c3, hipStream_t s) {
  return c3.colstats ? launch_wconv3s<true>(c3, s) : launch_wconv3s<false>(c3, s);
}

}  // namespace

hipError_t udt_lean_launch_gemm(int cfg, const void* lparams, int smem, int G, int geglu, int ln, hipStream_t s) {
  const lg::LParams& lp = *static_cast<const lg::LParams*>(lparams);
  const bool fp8 = lp.a_scale != nullptr, emit = lp.q8_out != nullptr;
  if (cfg == 7 && emit && !fp8 && !geglu && ln && lp.K == 320) return launch_rowres<5, false, true>(lp, G, s);
  if (fp8 || emit) return cfg == 1 ? launch_lean_mx8(lp, smem, G, geglu != 0, ln != 0, fp8, emit, s) : hipErrorInvalidValue;
  switch (cfg) {
    case 1: return launch_lean<4, 2, 2, 2, 2, 2>(lp, smem, G, geglu != 0, ln != 0, s);
    case 6: return launch_lean<8, 2, 4, 4, 2, 2, 1>(lp, smem, G, geglu != 0, ln != 0, s);
    case 5: return geglu ? hipErrorInvalidValue : launch_lean<4, 4, 1, 1, 5, 2>(lp, smem, G, false, ln != 0, s);
    case 7:                                               // rowres.h: LayerNorm-folded, K = 320, rows resident in registers
      if (!ln || lp.K != 320) return hipErrorInvalidValue;
      return geglu ? launch_rowres<5, true>(lp, G, s) : launch_rowres<5, false>(lp, G, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t udt_lean_launch_conv3(const void* c3params, hipStream_t s) {
  const lg::C3Params& c3 = *static_cast<const lg::C3Params*>(c3params);
  return c3.geo == 3 ? launch_wconv3(c3, s) : c3.geo == 2 ? launch_lconv3<16, 8, true>(c3, s)
         : c3.geo == 1 ? launch_lconv3<8, 8, false>(c3, s) : launch_lconv3<16, 8, false>(c3, s);
}
