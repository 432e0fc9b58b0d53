// rowres.h — ROW-RESIDENT LayerNorm-folded GEMM for the shallow-K projections of the UNet's 64 x 64 level (round 4).
//
// The GEGLU and q|k|v projections of the first level multiply 32768 rows (8 samples x 4096 tokens) by K = 320: a 256 x 256 output
// tile has FIVE K-tiles, so the tiled kernels (lean.h) spend most of a workgroup's life outside the MFMA loop — operand latency
// in front of it, the LayerNorm / GELU epilogue behind it — with nothing co-resident to overlap (profiles/r04_gemm_shapes.txt:
// 32768 x 2560 x 320 GEGLU 88 us = 607 TF/s, 24 % of the matrix peak; its epilogue alone is ~25 VALU operations per output).
// Here the loop order is turned around (the flash-attention order): a workgroup OWNS 256 rows and keeps their A fragments in
// REGISTERS for the whole launch — a wave holds its 64 rows x 320 K as 40 bf16x8 fragments (160 VGPRs, loaded once, straight
// from global memory; the LayerNorm row statistics come from the same registers) — and walks the weight matrix in 64-row chunks
// that stream through a three-deep LDS ring by LDS-DMA (40 KiB per chunk, the ONLY LDS traffic of the main loop: 0.5 KiB per MFMA
// against 0.75 of the 256 x 256 tile and 1.5 of the 128 x 128 one).  The accumulators of a chunk (64 rows x 64 columns per wave)
// ping-pong between two register sets, so the epilogue of chunk j - 1 (LayerNorm fold, bias, GEGLU, bf16, 16-byte stores) is
// issued BETWEEN the MFMAs of chunk j: one wave per SIMD, 512 registers, no co-resident workgroup needed to hide it.
//   * one barrier per chunk (all waves read the same weight chunk); the DMA for chunk j + 2 is issued right behind it
//   * bias / LayerNorm column constants of the workgroup's column range are copied to LDS once (<= 1280 columns)
//   * LN(x) W^T = rstd_m (x W'^T - mean_m s) + c as in lean.h; GEGLU pairs [32 x | 32 gate] per 64-row weight block
// Replaces lgemm_kernel<...LN> (same packed weights; the fp32 LayerNorm fold is factored as a_m acc + (b_m s_n + c_n), so results
// agree to fp32 rounding in front of the single bf16 rounding) for `ff.net[0]` (GEGLU, reference attention.py:53-66) and `attn1`'s
// q|k|v (attention.py:310-339) where K == 320.
#pragma once

// cost attribution (tools/rowres_cost_attribution.py): a COMPILE-TIME mask (-DUDT_MEASURE -DRR_DBG_MASK=<bits>: 0 no weight DMA
// after the first chunks, 1 no MFMA, 2 no epilogue math, 3 no A load, 4 no stores, 5 no fragment reads) — run-time switches put a
// branch around every MFMA and measured the branches
#if defined(UDT_MEASURE) && defined(RR_DBG_MASK)
#define RR_DBG(bit) ((((RR_DBG_MASK) >> (bit)) & 1) != 0)
#else
#define RR_DBG(bit) false
#endif

namespace rr {

using g8::wait_vm;
using g8::buf_lds16;

constexpr int RR_NRING = 3;
constexpr int rr_max_chunks(bool geglu) { return geglu ? 20 : 16; }   // column constants in LDS: chunks x 64 columns x (bias, s) fp32

template <int KT, bool GEGLU>
struct RGeo {
  static constexpr int CH_BYTES = KT * 64 * ROW_BYTES;   // one weight chunk: [K-tile][64 rows][128 B]
  static constexpr int STG_BYTES = GEGLU ? 64 * 64 : 64 * 128;     // per-wave staging: 64 rows x 64 B (GEGLU) / 128 B (plain)
  static constexpr int CST_BYTES = rr_max_chunks(GEGLU) * 64 * 2 * 4;
  static constexpr int SMEM = RR_NRING * CH_BYTES + 4 * STG_BYTES + CST_BYTES;
  static_assert(SMEM <= 160 * 1024, "one workgroup per CU");
};

// EMIT (plain epilogue only, round 5): the result ALSO as an MX8 activation (common.h) — q and k of a q|k|v projection with block
// scales along the head dimension, the v third (columns >= q8_fixed_col) with the fixed multiplier — for the e4m3 self-attention of
// BASELINE config #5 at the 64 x 64 level; `out` may then be null
template <int KT, bool GEGLU, bool EMIT = false>
__global__ void __launch_bounds__(256, 1) rgemm_kernel(const lg::LParams p) {
  static_assert(!(GEGLU && EMIT), "the emitting epilogue is the plain one");
  using Geo = RGeo<KT, GEGLU>;
  constexpr int KS = KT * 4;                             // 16-element K steps
  constexpr int CH_BYTES = Geo::CH_BYTES;
  constexpr int PPW = KT * 2;                            // 1-KiB DMA pieces per wave and chunk (KT * 8 pieces over 4 waves)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem;
  char* const cst = smem + RR_NRING * CH_BYTES + 4 * Geo::STG_BYTES;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int hi = lane >> 5;
  const int l3 = lane >> 3;
  const int pslot = lane & 7;
  const int swz = (l31 >> 1) & 7;
  char* const stg = smem + RR_NRING * CH_BYTES + wave * Geo::STG_BYTES;

  const int unit = range_index(blockIdx.x, p.G);
  if (unit >= p.tiles) return;
  const int tile_m = unit / p.tiles_n;                   // tiles_n = column splits; kt_per = chunks per split
  const int split = unit - tile_m * p.tiles_n;
  const int c0 = split * p.kt_per;
  int c1 = c0 + p.kt_per;
  const int nchunks = p.N >> 6;
  if (c1 > nchunks) c1 = nchunks;
  const int m0 = tile_m * 256 + wave * 64;

  // ---- weight chunk DMA: piece pc = wave + 4 i -> K-tile pc >> 3, rows (pc & 7) * 8 .. + 7 of the chunk
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w), 0, p.w_bytes, 0x00020000);
  unsigned w_voff[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pc = wave + 4 * i;
    const int row = (pc & 7) * 8 + l3;
    const int koff = (pc >> 3) * 64 + (pslot ^ ((row >> 1) & 7)) * 8;
    w_voff[i] = (unsigned)(((long long)row * p.ldw + koff) * 2);
  }
  const int chunk_step = 64 * p.ldw * 2;                 // bytes between chunks
  auto stage = [&](int c) {
    char* dst = ring + ((c - c0) % RR_NRING) * CH_BYTES;
#pragma unroll
    for (int i = 0; i < PPW; ++i) buf_lds16(rsrc_w, dst + (wave + 4 * i) * 1024, w_voff[i], c * chunk_step);
  };
  stage(c0);
  if (c0 + 1 < c1) stage(c0 + 1);

  // ---- EMIT: descriptors and lane offsets of the epilogue's buffer stores (see epilogue_store)
  __amdgpu_buffer_rsrc_t rs_q, rs_s, rs_o;
  int vq = 0, vs = 0, vo = 0;
  if constexpr (EMIT) {
    rs_q = __builtin_amdgcn_make_buffer_rsrc(p.q8_out, 0, (unsigned)(p.M * p.ld_q8), 0x00020000);
    rs_s = __builtin_amdgcn_make_buffer_rsrc(p.q8_scale, 0, (unsigned)(((p.N + 127) >> 7) * p.M * 4), 0x00020000);
    rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out ? (unsigned)(p.M * p.ldo * 2) : 0u, 0x00020000);
    vq = (lane >> 3) * p.ld_q8 + (lane & 7) * 8;
    vs = (lane >> 3) * 4 + ((lane & 7) >> 2);
    vo = ((lane >> 3) * p.ldo + (lane & 7) * 8) * 2;
  }

  // ---- column constants of this workgroup's range -> LDS ([chunk][0: bias | 1: s][64])
  {
    const int ncol = (c1 - c0) * 64;
    for (int i = tid; i < ncol; i += 256) {
      const int n = c0 * 64 + i;
      float* d = reinterpret_cast<float*>(cst) + (i >> 6) * 128 + (i & 63);
      // GEGLU: the bias carries the operand scale of geglu_scaled (columns [0, 32) of a chunk are x, [32, 64) the gate); the
      // LayerNorm column sums are multiplied by the row's (scaled) a_add
      const float bsc = GEGLU ? ((i & 32) ? GEGLU_GS : GEGLU_XS) : 1.0f;
      d[0] = p.bias ? p.bias[n] * bsc : 0.f;
      d[64] = p.ln_s[n];
    }
  }

  // ---- this wave's rows: A fragments (B operand of out^T = W x^T: 8 consecutive k at 16 ks + 8 hi of row l31) + LN statistics
  bf16x8_t xf[KS][2];
  float a_mul[2], a_add[2];                              // out = a_mul * acc + a_add * s_n + c_n
  float g_mul[2], g_add[2];                              // GEGLU: the gate's multipliers (a_* then belong to x), both with the
                                                         // operand scales of geglu_scaled folded in
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int m = m0 + tm * 32 + l31;
    const uint16_t* src = p.a + (long long)(m < p.M ? m : 0) * p.lda + hi * 8;
    float rs = 0.f, rq = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (!RR_DBG(3)) v = *reinterpret_cast<const u32x4*>(src + ks * 16);
      if (m >= p.M) v = u32x4{0u, 0u, 0u, 0u};
      xf[ks][tm] = __builtin_bit_cast(bf16x8_t, v);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        rs = lg::dot2_bf16(v[j], 0x3f803f80u, rs);
        rq = lg::dot2_bf16(v[j], v[j], rq);
      }
    }
    const float s = xor32_sum(rs), q = xor32_sum(rq);
    const float mean = s / (float)p.K;
    const float var = fmaxf(q / (float)p.K - mean * mean, 0.f);
    const float rstd = __builtin_amdgcn_rsqf(var + p.ln_eps);
    a_mul[tm] = rstd * p.alpha * (GEGLU ? GEGLU_XS : 1.0f);
    a_add[tm] = -mean * a_mul[tm];
    g_mul[tm] = rstd * p.alpha * GEGLU_GS;
    g_add[tm] = -mean * g_mul[tm];
  }

  f32x16 acc[2][2][2];                                   // [ping-pong][tm][tn]
  auto compute = [&](int c, f32x16 (&ac)[2][2]) {
    const char* cb = ring + ((c - c0) % RR_NRING) * CH_BYTES;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) ac[tm][tn][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const char* kb = cb + (ks >> 2) * (64 * ROW_BYTES) + l31 * ROW_BYTES + ((((ks & 3) * 2 + hi) ^ swz) << 4);
      const bf16x8_t w0 = lds_read_frag(kb);
      const bf16x8_t w1 = lds_read_frag(kb + 32 * ROW_BYTES);
      ac[0][0] = mfma32(w0, xf[ks][0], ac[0][0]);
      ac[1][0] = mfma32(w0, xf[ks][1], ac[1][0]);
      ac[0][1] = mfma32(w1, xf[ks][0], ac[0][1]);
      ac[1][1] = mfma32(w1, xf[ks][1], ac[1][1]);
    }
  };
  // acc[tm][tn][q * 4 + r] = out[row tm * 32 + l31][column tn * 32 + q * 8 + hi * 4 + r] of the chunk
  auto epilogue_math = [&](int c, const f32x16 (&ac)[2][2]) {
    const float* cc = reinterpret_cast<const float*>(cst) + (c - c0) * 128;
    if constexpr (GEGLU) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = q * 8 + hi * 4;
        const f32x4 bx = *reinterpret_cast<const f32x4*>(cc + col), bg = *reinterpret_cast<const f32x4*>(cc + 32 + col);
        const f32x4 sx = *reinterpret_cast<const f32x4*>(cc + 64 + col), sg = *reinterpret_cast<const f32x4*>(cc + 96 + col);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          const int row = tm * 32 + l31;
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x = a_mul[tm] * ac[tm][0][q * 4 + r] + (a_add[tm] * sx[r] + bx[r]);
            const float g = g_mul[tm] * ac[tm][1][q * 4 + r] + (g_add[tm] * sg[r] + bg[r]);
            o[r] = geglu_scaled(x, g);
          }
          u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
          *reinterpret_cast<u32x2*>(stg + row * 64 + ((q ^ (row & 3)) << 4) + hi * 8) = pk;
        }
      }
    } else {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = tn * 32 + q * 8 + hi * 4;
            const int row = tm * 32 + l31;
            const f32x4 b = *reinterpret_cast<const f32x4*>(cc + col), sn = *reinterpret_cast<const f32x4*>(cc + 64 + col);
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = a_mul[tm] * ac[tm][tn][q * 4 + r] + (a_add[tm] * sn[r] + b[r]);
            u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            // 16-byte slot (tn * 4 + q) of the row, XOR-swizzled by the row; the two half-waves fill its two halves
            *reinterpret_cast<u32x2*>(stg + row * 128 + (((tn * 4 + q) ^ (row & 7)) << 4) + hi * 8) = pk;
          }
    }
  };
  // staged bf16 rows -> global, 16 bytes per lane (a wave's LDS operations execute in order: no barrier around its own block)
  // (always_inline: with the emitting body hipcc left this lambda as a CALL — its closure, the kernel arguments included, then lives
  // in scratch and every field is reloaded through the vmcnt queue)
  auto epilogue_store = [&](int c) __attribute__((always_inline)) {
    if constexpr (GEGLU) {
      const int ch = lane & 3;
      uint16_t* const ob = p.out + c * 32 + ch * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 16 + (lane >> 2);
        const int m = m0 + row;
        const u32x4 v = *reinterpret_cast<const u32x4*>(stg + row * 64 + ((ch ^ (row & 3)) << 4));
        if (m < p.M && !RR_DBG(4)) *reinterpret_cast<u32x4*>(ob + (long long)m * p.ldo) = v;
      }
    } else {
      const int c8 = lane & 7;
      if constexpr (EMIT) {
        // Buffer stores: the descriptors sit in SGPRs, the lane part of every address in ONE VGPR each (e4m3 bytes, scale bytes,
        // bf16 result), the row / chunk part in an SGPR — 64-bit per-row pointers hoisted out of the chunk loop do not fit next
        // to the resident A fragments (they landed in scratch and their reloads in the vmcnt queue: 107 us instead of 41).
        // The 4 lanes of an aligned quad (c8 = 0..3 / 4..7) hold one row's 32-column block; a chunk is either block-scaled or,
        // from q8_fixed_col on (a multiple of 64), written with the fixed multiplier.
        const bool fixed = c * 64 >= p.q8_fixed_col;
        const int sq0 = m0 * p.ld_q8 + c * 64, ss0 = (((c >> 1) * p.M + m0) << 2) + ((2 * c) & 3), so0 = (m0 * p.ldo + c * 64) * 2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = i * 8 + (lane >> 3);
          const u32x4 v = *reinterpret_cast<const u32x4*>(stg + row * 128 + ((c8 ^ (row & 7)) << 4));
          const float f[8] = {bf16_lo(v[0]), bf16_hi(v[0]), bf16_lo(v[1]), bf16_hi(v[1]), bf16_lo(v[2]), bf16_hi(v[2]), bf16_lo(v[3]), bf16_hi(v[3])};
          uint32_t sb = 127u;
          const u32x2 q8 = fixed ? e4m3_fixed_row8(f, p.q8_fixed_mul) : mx8_quant_row8(f, sb);
          if (m0 + row < p.M) {
            __builtin_amdgcn_raw_buffer_store_b64(q8, rs_q, vq, sq0 + i * 8 * p.ld_q8, 0);
            if ((c8 & 3) == 0) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)sb, rs_s, vs, ss0 + i * 32, 0);
            if (p.out) __builtin_amdgcn_raw_buffer_store_b128(v, rs_o, vo, so0 + i * 16 * p.ldo, 0);
          }
        }
      } else {
        uint16_t* const ob = p.out + c * 64 + c8 * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = i * 8 + (lane >> 3);
          const int m = m0 + row;
          const u32x4 v = *reinterpret_cast<const u32x4*>(stg + row * 128 + ((c8 ^ (row & 7)) << 4));
          if (m < p.M && !RR_DBG(4)) *reinterpret_cast<u32x4*>(ob + (long long)m * p.ldo) = v;
        }
      }
    }
  };

  // ---- main loop: chunk j's MFMAs with chunk j - 1's epilogue between them.  Left alone, hipcc keeps the 80 MFMAs of a chunk in
  // one cluster and the epilogue behind it (one large sched_group_barrier pipeline over the whole chunk was not honoured either, and
  // took the compiler a minute).  The work is therefore cut into 16 UNITS fenced by sched_barrier(0): five MFMAs of chunk j (the
  // fragment reads one K step ahead) + one sixteenth of chunk j - 1's epilogue (GEGLU: two outputs of one (q, tm) group, ~45 VALU
  // instructions; plain: one (tm, tn, q) group of four), and inside a unit a small pipeline pins one MFMA + its share of VALU
  // work per 32-cycle matrix slot.
  constexpr int NUNIT = 16, MPU = (KS * 4) / NUNIT;      // MFMAs per unit
  static_assert(MPU * NUNIT == KS * 4, "units tile the chunk's MFMAs");
  auto fused = [&](int cj, f32x16 (&an)[2][2], const f32x16 (&ao)[2][2]) {
    const char* cb = ring + ((cj - c0) % RR_NRING) * CH_BYTES + l31 * ROW_BYTES;
    const float* cc = reinterpret_cast<const float*>(cst) + (cj - 1 - c0) * 128;
    int foff[4];
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) foff[k4] = ((k4 * 2 + hi) ^ swz) << 4;
    bf16x8_t wf[2][2];
    wf[0][0] = lds_read_frag(cb + foff[0]);
    wf[0][1] = lds_read_frag(cb + 32 * ROW_BYTES + foff[0]);
    f32x4 k0, k1, k2, k3;                                // column constants of the current group
    float o[4];
#pragma unroll
    for (int u = 0; u < NUNIT; ++u) {
      // -- MFMAs 5u .. 5u + 4 of the chunk: j = 4 ks + w, w -> (tm, tn) = (w & 1, w >> 1)
#pragma unroll
      for (int i = 0; i < MPU; ++i) {
        const int j = u * MPU + i, ks = j >> 2, w = j & 3;
        if (w == 0 && ks + 1 < KS && !RR_DBG(5)) {
          const char* kb = cb + ((ks + 1) >> 2) * (64 * ROW_BYTES) + foff[(ks + 1) & 3];
          wf[(ks + 1) & 1][0] = lds_read_frag(kb);
          wf[(ks + 1) & 1][1] = lds_read_frag(kb + 32 * ROW_BYTES);
        }
        f32x16& d = an[w & 1][w >> 1];
        if (RR_DBG(1)) {
          if (ks == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
          }
          asm volatile("" : "+v"(wf[ks & 1][w >> 1]));
        } else if (ks == 0) {
          f32x16 z;
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = 0.f;
          d = mfma32(wf[0][w >> 1], xf[0][w & 1], z);
        } else {
          d = mfma32(wf[ks & 1][w >> 1], xf[ks][w & 1], d);
        }
      }
      // -- epilogue slice u of chunk cj - 1
      if (RR_DBG(2)) {
      } else if constexpr (GEGLU) {
        const int q = u >> 2, tm = (u >> 1) & 1, half = u & 1;
        if ((u & 3) == 0) {
          const int col = q * 8 + hi * 4;
          k0 = *reinterpret_cast<const f32x4*>(cc + col);          // bias x
          k1 = *reinterpret_cast<const f32x4*>(cc + 32 + col);     // bias gate
          k2 = *reinterpret_cast<const f32x4*>(cc + 64 + col);     // s x
          k3 = *reinterpret_cast<const f32x4*>(cc + 96 + col);     // s gate
        }
#pragma unroll
        for (int rr2 = 0; rr2 < 2; ++rr2) {
          const int r = half * 2 + rr2;
          const float x = a_mul[tm] * ao[tm][0][q * 4 + r] + (a_add[tm] * k2[r] + k0[r]);
          const float g = g_mul[tm] * ao[tm][1][q * 4 + r] + (g_add[tm] * k3[r] + k1[r]);
          o[r] = geglu_scaled(x, g);
        }
        if (half == 1) {
          const int row = tm * 32 + l31;
          u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
          *reinterpret_cast<u32x2*>(stg + row * 64 + ((q ^ (row & 3)) << 4) + hi * 8) = pk;
        }
      } else {
        const int tm = u >> 3, tn = (u >> 2) & 1, q = u & 3;
        const int col = tn * 32 + q * 8 + hi * 4, row = tm * 32 + l31;
        k0 = *reinterpret_cast<const f32x4*>(cc + col);
        k2 = *reinterpret_cast<const f32x4*>(cc + 64 + col);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = a_mul[tm] * ao[tm][tn][q * 4 + r] + (a_add[tm] * k2[r] + k0[r]);
        u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
        *reinterpret_cast<u32x2*>(stg + row * 128 + (((tn * 4 + q) ^ (row & 7)) << 4) + hi * 8) = pk;
      }
#pragma unroll
      for (int i = 0; i < MPU; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, GEGLU ? 8 : 3, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    epilogue_store(cj - 1);
  };
  int c = c0;
  auto top = [&](int cj) {
    // chunk cj landed: behind it at most the DMA of chunk cj + 1 and the (<= 8) stores of the last epilogue are in flight; waiting
    // for all but the PPW youngest operations covers it (vmcnt retires in order and counts stores too)
    // (EMIT: an epilogue issues 16 stores — 8 of e4m3 bytes, 8 of scale bytes — or 24 with the bf16 result: with the plain count the
    // wait would reach into the stores issued a moment ago and expose their latency at every chunk)
    // (... but only a wave whose 64 rows all lie inside M issues those 16: the stores sit under the row predicate, hipcc branches around
    // fully masked VMEM, and a wave with < 8 stores per epilogue would let the wider count reach into chunk cj's own DMA pieces — the
    // other waves would then read a weight chunk that has not landed.  Ragged waves take the count that is safe for any store count)
    if (cj + 1 >= c1) wait_vm<0>();
    else if (EMIT && m0 + 64 <= p.M) wait_vm<PPW + 16>();
    else wait_vm<PPW>();
    lg::raw_barrier();                                   // every wave is past chunk cj - 1: its ring slot takes chunk cj + 2
    if (cj + 2 < c1 && !RR_DBG(0)) stage(cj + 2);
  };
  top(c);
  compute(c, acc[0]);
  for (++c; c + 1 < c1; c += 2) {
    top(c);
    fused(c, acc[1], acc[0]);
    top(c + 1);
    fused(c + 1, acc[0], acc[1]);
  }
  if (c < c1) {
    top(c);
    fused(c, acc[1], acc[0]);
    epilogue_math(c, acc[1]);
    epilogue_store(c);
  } else {
    epilogue_math(c - 1, acc[0]);
    epilogue_store(c - 1);
  }
}

}  // namespace rr
