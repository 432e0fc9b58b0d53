// gemm8.h — the 8-wave, deep-pipelined, persistent GEMM / implicit-conv kernel (included by gemm.hip).
//
// One workgroup = 512 threads = 8 waves owns a BM x BN output tile (256x128: waves 4x2, 64x64 each; or 256x160:
// waves 8x1, 32x160 each — the latter tiles N = 320 without waste).  One workgroup per CU (G <= #CUs), all
// co-resident, each walking an equal contiguous share of the (tile, K-tile) iteration space ("stream-K").
//
// Pipeline: a ring of THREE LDS stages (each BM+BN rows x 128 B, XOR-swizzled, filled by 16-byte LDS-DMA).  Loads run
// two K-tiles ahead of the MFMAs and stay in flight across the workgroup barrier: the loop uses a raw s_barrier and a
// COUNTED `s_waitcnt vmcnt(loads of one K-tile)` — never vmcnt(0) — so the only thing a K-tile waits for is data that
// was requested two tiles ago (cdna_hip_programming.md §5 "Pipelining across barriers", T3+T4).  Two waves per SIMD
// let one wave's ds_read / address work overlap the other's MFMAs.
//
// Stream-K finishing happens inside the launch: a workgroup whose range starts in the middle of a tile computes that
// head segment FIRST, parks the fp32 accumulators in its slab and raises its flag (agent-scope release); the workgroup
// that owns the START of the tile reaches it LAST in its range, polls the partner flags (relaxed poll, one agent-scope
// acquire), adds the slabs and runs the fused epilogue.  Dependencies only point at work a partner does first, so
// there is no wait chain; flags are reset by their consumer, the workspace is zero-initialised once by the caller.
#pragma once

namespace g8 {

constexpr int NSTAGE = 3;
constexpr int NTHREADS = 512;
constexpr int SPIN_LIMIT = 1 << 24;

struct Params {
  GemmParams g;          // operand / epilogue description shared with the 4-wave kernel
  int* flags;            // [G] slab-ready flags (0 / 1), zero between launches
  int* err;              // [1] set to 1 if a spin timed out
  float* slab_base;      // [G][BM*BN] fp32
  unsigned a_bytes, w_bytes;   // buffer-descriptor extents (per batch element) of A and W
};

template <int TM, int TN, bool TRANS>
UDT_DEVINL void epilogue8(const GemmParams& p, f32x16 (&acc)[TM][TN], int m0, int n0, int batch, int row0, int col0,
                          int lane) {
  const int l31 = lane & 31;
  const int hi = lane >> 5;
  const int flags = p.flags;
  if constexpr (TRANS) {
    uint16_t* outT = reinterpret_cast<uint16_t*>(p.out);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + col0 + tn * 32 + l31;
        const float bias = (p.bias != nullptr && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = m0 + row0 + tm * 32 + q * 8 + hi * 4;
          if (m < p.M && n < p.N) {
            const int b = m / p.rows_per_batch;
            const int tok = m - b * p.rows_per_batch;
            const float v0 = acc[tm][tn][q * 4 + 0] * p.alpha + bias;
            const float v1 = acc[tm][tn][q * 4 + 1] * p.alpha + bias;
            const float v2 = acc[tm][tn][q * 4 + 2] * p.alpha + bias;
            const float v3 = acc[tm][tn][q * 4 + 3] * p.alpha + bias;
            u32x2 pk = {pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
            const long long off = ((long long)b * p.N + n) * p.rows_per_batch + tok;
            *reinterpret_cast<u32x2*>(outT + off) = pk;
          }
        }
      }
  } else {
    const uint16_t* __restrict__ R = p.res ? (p.res + (long long)batch * p.sR) : nullptr;
    if (flags & UDT_GEMM_GEGLU) {
      if constexpr (TN == 2) {
        uint16_t* out = reinterpret_cast<uint16_t*>(p.out) + (long long)batch * p.sO;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          const int m = m0 + row0 + tm * 32 + l31;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int nx = n0 + col0 + q * 8 + hi * 4;
            const int no = ((n0 + col0) >> 1) + q * 8 + hi * 4;
            if (m < p.M && nx < p.N) {
              float o[4];
              f32x4 bx = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
              if (p.bias) {
                bx = *reinterpret_cast<const f32x4*>(p.bias + nx);
                bg = *reinterpret_cast<const f32x4*>(p.bias + nx + 32);
              }
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float x = acc[tm][0][q * 4 + r] * p.alpha + bx[r];
                const float gt = acc[tm][1][q * 4 + r] * p.alpha + bg[r];
                o[r] = x * gelu_erf_f(gt);
              }
              u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
              *reinterpret_cast<u32x2*>(out + (long long)m * p.ldo + no) = pk;
            }
          }
        }
      }
      return;
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int m = m0 + row0 + tm * 32 + l31;
      const int b = (p.rowvec != nullptr) ? (m / p.rows_per_batch) : 0;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + col0 + tn * 32 + q * 8 + hi * 4;
          if (m < p.M && n < p.N) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[tm][tn][q * 4 + r] * p.alpha;
            if (p.bias) {
              const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += bv[r];
            }
            if (p.rowvec) {
              const f32x4 rv = *reinterpret_cast<const f32x4*>(p.rowvec + (long long)b * p.ldrv + n);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += rv[r];
            }
            if (R) {
              const u32x2 rr = *reinterpret_cast<const u32x2*>(R + (long long)m * p.ldr + n);
              v[0] += bf16_lo(rr[0]);
              v[1] += bf16_hi(rr[0]);
              v[2] += bf16_lo(rr[1]);
              v[3] += bf16_hi(rr[1]);
            }
            if (flags & UDT_GEMM_RELU) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (flags & UDT_GEMM_SILU_OUT) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
            }
            if (flags & UDT_GEMM_OUT_F32) {
              float* out = reinterpret_cast<float*>(p.out) + (long long)batch * p.sO;
              f32x4 ov = {v[0], v[1], v[2], v[3]};
              *reinterpret_cast<f32x4*>(out + (long long)m * p.ldo + n) = ov;
            } else {
              uint16_t* out = reinterpret_cast<uint16_t*>(p.out) + (long long)batch * p.sO;
              u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
              *reinterpret_cast<u32x2*>(out + (long long)m * p.ldo + n) = pk;
            }
          }
        }
    }
  }
}

// ---- column statistics of the output (GroupNorm partial sums for the NEXT layer) -------------------------------------
// udt_gemm_desc.colstats: fp32 [slots][N][2] = per-(row slot, column) (sum, sum of squares) of the stored values.  A slot
// is one wave's row block (64 rows of the 256x128 tile, 32 rows of the 256x160 tile), written exactly once by the
// workgroup that finishes the tile: plain stores, no atomics, so the statistics are deterministic.
UDT_DEVINL void colstat_acc8(const u32x4 v, float (&s)[8], float (&q)[8]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = bf16_lo(v[j]), b = bf16_hi(v[j]);
    s[2 * j] += a; q[2 * j] += a * a;
    s[2 * j + 1] += b; q[2 * j + 1] += b * b;
  }
}
// rows layout (8 lanes x 16 B cover one 128-byte row; lane>>3 = row within a group of 8): sum over the 8 row lanes
// (lane bits 3, 4, 5) with row_ror:8 + the row / half swaps; the lanes of row group 0 write their 8 columns
UDT_DEVINL void colstat_emit_rows(float (&s)[8], float (&q)[8], int lane, float* dst, int ncols) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s[j] = xor32_sum(xor16_sum(dpp_add<0x128>(s[j])));
    q[j] = xor32_sum(xor16_sum(dpp_add<0x128>(q[j])));
  }
  const int col = (lane & 7) * 8;
  if (lane < 8 && col < ncols) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      f32x4 o = {s[j], q[j], s[j + 1], q[j + 1]};
      *reinterpret_cast<f32x4*>(dst + (col + j) * 2) = o;
    }
  }
}
// accumulator layout (lane = one row, 16 (q, r) columns of a 32-column MFMA tile): sum over the 32 row lanes of each
// half-wave; lane 0 / lane 32 write the 16 columns of their half
// NOT inlined: inlined into the 512-thread, 256-VGPR convolution kernel this code produced launches that died with
// HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION in one or the other (GN, STATS) variant depending on unrelated code
// layout (ROCm 7.2 hipcc; tools/probes/variants.py); behind a call boundary all variants are correct.
__device__ __attribute__((noinline)) void colstat_emit_acc(float (&s16)[16], float (&q16)[16], int lane, float* dst_tile, int ncols) {
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    s16[j] = xor16_sum(row16_sum(s16[j]));
    q16[j] = xor16_sum(row16_sum(q16[j]));
  }
  if ((lane & 31) == 0) {
    const int hi = lane >> 5;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int col = qd * 8 + hi * 4;             // columns of (q = qd, r = 0..3)
      if (col < ncols) {
        f32x4 o0 = {s16[qd * 4 + 0], q16[qd * 4 + 0], s16[qd * 4 + 1], q16[qd * 4 + 1]};
        f32x4 o1 = {s16[qd * 4 + 2], q16[qd * 4 + 2], s16[qd * 4 + 3], q16[qd * 4 + 3]};
        *reinterpret_cast<f32x4*>(dst_tile + col * 2) = o0;
        *reinterpret_cast<f32x4*>(dst_tile + col * 2 + 4) = o1;
      }
    }
  }
}

// ---- row-coalesced epilogue (bf16 output, TN == 2, not transposed) ----------------------------------------------
// The MFMA accumulator layout gives a lane 4 consecutive columns of ONE row, so a direct store instruction touches 32
// rows with 16 bytes each: 512 partial-line write requests per wave and tile, and the launch becomes L2-request
// bound (a K=64 GEMM wrote its output at 2.4 TB/s).  Instead every wave transposes its 64-row block through 8 KiB of
// LDS of its own (XOR-swizzled 16-byte chunks) and stores 16 bytes per lane with 8 lanes covering one full 128-byte
// row: whole cache lines, half the store instructions.  Residual rows are fetched the same way (whole lines) and
// added in fp32 BEFORE the single bf16 rounding.  `wlds` = this wave's 8 KiB scratch (free ring stage).
// `mof(row)` maps a wave-local row (0 .. TM*32-1) to the global output row, or -1 (GEMM: m0 + row0 + row; the
// patch-staged convolution: the NHWC pixel of that tile position); `nw` = first output column of the wave.
// `stats`: this wave block's row of p.colstats (+ 2 * nw), or nullptr
template <int TM, class MOF>
UDT_DEVINL void epilogue8_rows(const GemmParams& p, f32x16 (&acc)[TM][2], MOF mof, int nw, int batch, int lane,
                               char* wlds, float* stats) {
  const int l31 = lane & 31;
  const int hi = lane >> 5;
  const int flags = p.flags;
  uint16_t* out = reinterpret_cast<uint16_t*>(p.out) + (long long)batch * p.sO;
  if (flags & UDT_GEMM_GEGLU) {
    // 32 output columns per wave: rows of 64 bytes, 4 chunks; lane -> (row = 16 i + lane/4, chunk = lane%4)
    const int nx0 = nw;
    const int no0 = nw >> 1;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int row = tm * 32 + l31;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nx = nx0 + q * 8 + hi * 4;
        f32x4 bx = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && nx < p.N && !UDT_DBG(p.flags, 24)) {
          bx = *reinterpret_cast<const f32x4*>(p.bias + nx);
          bg = *reinterpret_cast<const f32x4*>(p.bias + nx + 32);
        }
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          o[r] = (acc[tm][0][q * 4 + r] * p.alpha + bx[r]) * gelu_erf_f(acc[tm][1][q * 4 + r] * p.alpha + bg[r]);
        u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
        *reinterpret_cast<u32x2*>(wlds + row * 64 + ((q ^ (row & 3)) << 4) + hi * 8) = pk;
      }
    }
#pragma unroll
    for (int i = 0; i < TM * 2; ++i) {
      const int row = i * 16 + (lane >> 2);
      const int ch = lane & 3;
      const u32x4 v = *reinterpret_cast<const u32x4*>(wlds + row * 64 + ((ch ^ (row & 3)) << 4));
      const long long m = mof(row);
      const int no = no0 + ch * 8;
      if (m >= 0 && (nx0 + ch * 8) < p.N && !UDT_DBG(p.flags, 26)) *reinterpret_cast<u32x4*>(out + m * p.ldo + no) = v;
    }
    return;
  }
  const uint16_t* __restrict__ R = (p.res && !UDT_DBG(p.flags, 25)) ? (p.res + (long long)batch * p.sR) : nullptr;
  if (R) {
    // residual block -> LDS in whole rows (8 lanes x 16 B = one 128-byte line), read back in accumulator order below
#pragma unroll
    for (int i = 0; i < TM * 4; ++i) {
      const int row = i * 8 + (lane >> 3);
      const int ch = lane & 7;
      const long long m = mof(row);
      const int n = nw + ch * 8;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (m >= 0 && n + 8 <= p.N) {
        v = *reinterpret_cast<const u32x4*>(R + m * p.ldr + n);
      } else if (m >= 0 && n < p.N) {                      // N % 8 == 4 tail: 4 columns
        const u32x2 h = *reinterpret_cast<const u32x2*>(R + m * p.ldr + n);
        v[0] = h[0];
        v[1] = h[1];
      }
      *reinterpret_cast<u32x4*>(wlds + row * 128 + ((ch ^ (row & 7)) << 4)) = v;
    }
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int row = tm * 32 + l31;
    const long long mo = mof(row);
    const int b = (p.rowvec != nullptr) ? (int)((mo >= 0 ? mo : 0) / p.rows_per_batch) : 0;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = nw + tn * 32 + q * 8 + hi * 4;
        char* cell = wlds + row * 128 + (((tn * 4 + q) ^ (row & 7)) << 4) + hi * 8;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[tm][tn][q * 4 + r] * p.alpha;
        if (n < p.N) {
          if (p.bias && !UDT_DBG(p.flags, 24)) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += bv[r];
          }
          if (p.rowvec && !UDT_DBG(p.flags, 24)) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(p.rowvec + (long long)b * p.ldrv + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += rv[r];
          }
        }
        if (R) {
          const u32x2 rr = *reinterpret_cast<const u32x2*>(cell);
          v[0] += bf16_lo(rr[0]);
          v[1] += bf16_hi(rr[0]);
          v[2] += bf16_lo(rr[1]);
          v[3] += bf16_hi(rr[1]);
        }
        if (flags & UDT_GEMM_RELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (flags & UDT_GEMM_SILU_OUT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
        }
        u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<u32x2*>(cell) = pk;
      }
  }
  float cs[8], cq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = cq[j] = 0.f;
#pragma unroll
  for (int i = 0; i < TM * 4; ++i) {
    const int row = i * 8 + (lane >> 3);
    const int ch = lane & 7;
    const u32x4 v = *reinterpret_cast<const u32x4*>(wlds + row * 128 + ((ch ^ (row & 7)) << 4));
    const long long m = mof(row);
    const int n = nw + ch * 8;
    if (m >= 0 && n + 8 <= p.N && !UDT_DBG(p.flags, 26)) {
      *reinterpret_cast<u32x4*>(out + m * p.ldo + n) = v;
    } else if (m >= 0 && n < p.N) {
      u32x2 h = {v[0], v[1]};
      *reinterpret_cast<u32x2*>(out + m * p.ldo + n) = h;
    }
    if (stats && m >= 0) colstat_acc8(v, cs, cq);
  }
  if (stats) colstat_emit_rows(cs, cq, lane, stats, p.N - nw);
}

// the same for the 256x160 configuration (TM = 1, TN = 5: a wave owns 32 rows x 160 columns = 320-byte rows).  The
// free ring stage holds 52 KiB, so the block goes through LDS in two passes of 16 rows (rows padded to 336 bytes:
// 84-dword stride spreads the 16 rows over the banks without a swizzle); 20 lanes cover one row, 3 rows per
// instruction.
// 20 lanes x 16 B cover one 320-byte row, 3 rows per instruction: column sums = own rows + the two partner lanes
template <int NC>
UDT_DEVINL void colstat_emit_rows16(float (&s)[8], float (&q)[8], int lane, float* dst, int ncols) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s[j] += __shfl(s[j], lane + NC) + __shfl(s[j], lane + 2 * NC);
    q[j] += __shfl(q[j], lane + NC) + __shfl(q[j], lane + 2 * NC);
  }
  if (lane < NC && lane * 8 < ncols) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      f32x4 o = {s[j], q[j], s[j + 1], q[j + 1]};
      *reinterpret_cast<f32x4*>(dst + (lane * 8 + j) * 2) = o;
    }
  }
}

template <int TN, class MOF>
UDT_DEVINL void epilogue8_rows16(const GemmParams& p, f32x16 (&acc)[1][TN], MOF mof, int nw, int batch, int lane,
                                 char* wlds, float* stats) {
  constexpr int NC = TN * 4;                   // 16-byte chunks per row
  constexpr int RS = TN * 64 + 16;             // padded row stride
  const int l31 = lane & 31;
  const int hi = lane >> 5;
  const int flags = p.flags;
  uint16_t* out = reinterpret_cast<uint16_t*>(p.out) + (long long)batch * p.sO;
  const uint16_t* __restrict__ R = (p.res && !UDT_DBG(p.flags, 25)) ? (p.res + (long long)batch * p.sR) : nullptr;
  const int rl = lane / NC;                    // 0..2 (lane 60..63: idle)
  const int ch = lane - rl * NC;
  float cs[8], cq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = cq[j] = 0.f;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const bool mine = (l31 >> 4) == pass;
    const int lrow = l31 & 15;
    if (R) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int row = i * 3 + rl;
        const long long m = mof(pass * 16 + (row < 16 ? row : 15));   // outside the branch: mof may read another lane
        if (rl < 3 && row < 16) {
          const int n = nw + ch * 8;
          u32x4 v = {0u, 0u, 0u, 0u};
          if (m >= 0 && n + 8 <= p.N) {
            v = *reinterpret_cast<const u32x4*>(R + m * p.ldr + n);
          } else if (m >= 0 && n < p.N) {
            const u32x2 h = *reinterpret_cast<const u32x2*>(R + m * p.ldr + n);
            v[0] = h[0];
            v[1] = h[1];
          }
          *reinterpret_cast<u32x4*>(wlds + row * RS + ch * 16) = v;
        }
      }
    }
    const long long mo = mof(l31);          // (all lanes: mof may be a cross-lane read)
    if (mine) {
      const int b = (p.rowvec != nullptr) ? (int)((mo >= 0 ? mo : 0) / p.rows_per_batch) : 0;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nw + tn * 32 + q * 8 + hi * 4;
          char* cell = wlds + lrow * RS + (tn * 4 + q) * 16 + hi * 8;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[0][tn][q * 4 + r] * p.alpha;
          if (n < p.N) {
            if (p.bias && !UDT_DBG(p.flags, 24)) {
              const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += bv[r];
            }
            if (p.rowvec && !UDT_DBG(p.flags, 24)) {
              const f32x4 rv = *reinterpret_cast<const f32x4*>(p.rowvec + (long long)b * p.ldrv + n);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += rv[r];
            }
          }
          if (R) {
            const u32x2 rr = *reinterpret_cast<const u32x2*>(cell);
            v[0] += bf16_lo(rr[0]);
            v[1] += bf16_hi(rr[0]);
            v[2] += bf16_lo(rr[1]);
            v[3] += bf16_hi(rr[1]);
          }
          if (flags & UDT_GEMM_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          }
          if (flags & UDT_GEMM_SILU_OUT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
          }
          u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          *reinterpret_cast<u32x2*>(cell) = pk;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int row = i * 3 + rl;
      const long long m = mof(pass * 16 + (row < 16 ? row : 15));
      if (rl < 3 && row < 16) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(wlds + row * RS + ch * 16);
        const int n = nw + ch * 8;
        if (m >= 0 && n + 8 <= p.N && !UDT_DBG(p.flags, 26)) {
          *reinterpret_cast<u32x4*>(out + m * p.ldo + n) = v;
        } else if (m >= 0 && n < p.N) {
          u32x2 h = {v[0], v[1]};
          *reinterpret_cast<u32x2*>(out + m * p.ldo + n) = h;
        }
        if (stats && m >= 0) colstat_acc8(v, cs, cq);
      }
    }
  }
  if (stats) colstat_emit_rows16<NC>(cs, cq, lane, stats, p.N - nw);
}

// ---- fast variants of the row-coalesced epilogues ----------------------------------------------------------------
// Measured on MI355X: the generic epilogues above spend ~9 us per tile on NOTHING BUT control flow — every 4-value
// group re-tests the runtime feature flags and its lane predicates, and every bias / time-embedding load sits behind
// such a branch, so the loads are issued one latency at a time (another ~8 us).  The fast variants are compiled per
// feature set (bias / per-sample row vector / residual), assume an INTERIOR tile (every row and column valid, one
// sample per wave block) and issue all their loads up front; anything else takes the generic path.
template <int TM, bool BIAS, bool ROWVEC, bool RES, class MOF>
UDT_DEVINL void epilogue8_rows_fast(const GemmParams& p, f32x16 (&acc)[TM][2], MOF mof, int nw, int batch, int lane,
                                    char* wlds, float* stats) {
  const int l31 = lane & 31;
  const int hi = lane >> 5;
  uint16_t* out = reinterpret_cast<uint16_t*>(p.out) + (long long)batch * p.sO;
  long long mrow[TM * 4];
#pragma unroll
  for (int i = 0; i < TM * 4; ++i) mrow[i] = mof(i * 8 + (lane >> 3));
  const int ch = lane & 7;
  if constexpr (RES) {
    const uint16_t* __restrict__ R = p.res + (long long)batch * p.sR;
    u32x4 rv[TM * 4];
#pragma unroll
    for (int i = 0; i < TM * 4; ++i) rv[i] = *reinterpret_cast<const u32x4*>(R + mrow[i] * p.ldr + nw + ch * 8);
#pragma unroll
    for (int i = 0; i < TM * 4; ++i) {
      const int row = i * 8 + (lane >> 3);
      *reinterpret_cast<u32x4*>(wlds + row * 128 + ((ch ^ (row & 7)) << 4)) = rv[i];
    }
  }
  f32x4 cv[2][4];
  int b = 0;
  if constexpr (ROWVEC) b = (int)(mof(0) / p.rows_per_batch);    // one sample per wave block (caller checked)
#pragma unroll
  for (int tn = 0; tn < 2; ++tn)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = nw + tn * 32 + q * 8 + hi * 4;
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      if constexpr (BIAS) c = *reinterpret_cast<const f32x4*>(p.bias + n);
      if constexpr (ROWVEC) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(p.rowvec + (long long)b * p.ldrv + n);
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] += r[k];
      }
      cv[tn][q] = c;
    }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int row = tm * 32 + l31;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        char* cell = wlds + row * 128 + (((tn * 4 + q) ^ (row & 7)) << 4) + hi * 8;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[tm][tn][q * 4 + r] * p.alpha + cv[tn][q][r];
        if constexpr (RES) {
          const u32x2 rr = *reinterpret_cast<const u32x2*>(cell);
          v[0] += bf16_lo(rr[0]);
          v[1] += bf16_hi(rr[0]);
          v[2] += bf16_lo(rr[1]);
          v[3] += bf16_hi(rr[1]);
        }
        u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<u32x2*>(cell) = pk;
      }
  }
  float cs[8], cq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = cq[j] = 0.f;
#pragma unroll
  for (int i = 0; i < TM * 4; ++i) {
    const int row = i * 8 + (lane >> 3);
    const u32x4 v = *reinterpret_cast<const u32x4*>(wlds + row * 128 + ((ch ^ (row & 7)) << 4));
    *reinterpret_cast<u32x4*>(out + mrow[i] * p.ldo + nw + ch * 8) = v;
    if (stats) colstat_acc8(v, cs, cq);
  }
  if (stats) colstat_emit_rows(cs, cq, lane, stats, 64);
}

// GEGLU, interior tile, bias present
template <int TM, class MOF>
UDT_DEVINL void epilogue8_geglu_fast(const GemmParams& p, f32x16 (&acc)[TM][2], MOF mof, int nw, int batch, int lane,
                                     char* wlds) {
  const int l31 = lane & 31;
  const int hi = lane >> 5;
  uint16_t* out = reinterpret_cast<uint16_t*>(p.out) + (long long)batch * p.sO;
  f32x4 bx[4], bg[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    bx[q] = *reinterpret_cast<const f32x4*>(p.bias + nw + q * 8 + hi * 4);
    bg[q] = *reinterpret_cast<const f32x4*>(p.bias + nw + q * 8 + hi * 4 + 32);
  }
  long long mrow[TM * 2];
#pragma unroll
  for (int i = 0; i < TM * 2; ++i) mrow[i] = mof(i * 16 + (lane >> 2));
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int row = tm * 32 + l31;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        o[r] = (acc[tm][0][q * 4 + r] * p.alpha + bx[q][r]) * gelu_erf_f(acc[tm][1][q * 4 + r] * p.alpha + bg[q][r]);
      u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
      *reinterpret_cast<u32x2*>(wlds + row * 64 + ((q ^ (row & 3)) << 4) + hi * 8) = pk;
    }
  }
  const int no0 = nw >> 1;
#pragma unroll
  for (int i = 0; i < TM * 2; ++i) {
    const int row = i * 16 + (lane >> 2);
    const int ch = lane & 3;
    const u32x4 v = *reinterpret_cast<const u32x4*>(wlds + row * 64 + ((ch ^ (row & 3)) << 4));
    *reinterpret_cast<u32x4*>(out + mrow[i] * p.ldo + no0 + ch * 8) = v;
  }
}

// 256x160 configuration (TM = 1): per-column constants (bias + row vector) staged once in LDS behind the 16-row block
template <int TN, bool BIAS, bool ROWVEC, bool RES, class MOF>
UDT_DEVINL void epilogue8_rows16_fast(const GemmParams& p, f32x16 (&acc)[1][TN], MOF mof, int nw, int batch, int lane,
                                      char* wlds, float* stats) {
  constexpr int NC = TN * 4;
  constexpr int RS = TN * 64 + 16;
  const int l31 = lane & 31;
  const int hi = lane >> 5;
  uint16_t* out = reinterpret_cast<uint16_t*>(p.out) + (long long)batch * p.sO;
  float* cvec = reinterpret_cast<float*>(wlds + 16 * RS);          // [TN * 32] floats
  int b = 0;
  if constexpr (ROWVEC) b = (int)(mof(0) / p.rows_per_batch);
  if constexpr (BIAS || ROWVEC) {
    if (lane < TN * 8) {
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      if constexpr (BIAS) c = *reinterpret_cast<const f32x4*>(p.bias + nw + lane * 4);
      if constexpr (ROWVEC) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(p.rowvec + (long long)b * p.ldrv + nw + lane * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] += r[k];
      }
      *reinterpret_cast<f32x4*>(cvec + lane * 4) = c;
    }
  }
  const int rl = lane / NC;
  const int ch = lane - rl * NC;
  long long mrow[2][6];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int row = i * 3 + rl;
      mrow[pass][i] = mof(pass * 16 + (row < 16 ? row : 15));
    }
  float cs[8], cq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = cq[j] = 0.f;
  u32x4 rv[2][6];
  if constexpr (RES) {
    const uint16_t* __restrict__ R = p.res + (long long)batch * p.sR;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass)
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int row = i * 3 + rl;
        if (rl < 3 && row < 16) rv[pass][i] = *reinterpret_cast<const u32x4*>(R + mrow[pass][i] * p.ldr + nw + ch * 8);
      }
  }
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const bool mine = (l31 >> 4) == pass;
    const int lrow = l31 & 15;
    if constexpr (RES) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int row = i * 3 + rl;
        if (rl < 3 && row < 16) *reinterpret_cast<u32x4*>(wlds + row * RS + ch * 16) = rv[pass][i];
      }
    }
    if (mine) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          char* cell = wlds + lrow * RS + (tn * 4 + q) * 16 + hi * 8;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[0][tn][q * 4 + r] * p.alpha;
          if constexpr (BIAS || ROWVEC) {
            const f32x4 c = *reinterpret_cast<const f32x4*>(cvec + tn * 32 + q * 8 + hi * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += c[r];
          }
          if constexpr (RES) {
            const u32x2 rr = *reinterpret_cast<const u32x2*>(cell);
            v[0] += bf16_lo(rr[0]);
            v[1] += bf16_hi(rr[0]);
            v[2] += bf16_lo(rr[1]);
            v[3] += bf16_hi(rr[1]);
          }
          u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          *reinterpret_cast<u32x2*>(cell) = pk;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int row = i * 3 + rl;
      if (rl < 3 && row < 16) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(wlds + row * RS + ch * 16);
        *reinterpret_cast<u32x4*>(out + mrow[pass][i] * p.ldo + nw + ch * 8) = v;
        if (stats) colstat_acc8(v, cs, cq);
      }
    }
  }
  if (stats) colstat_emit_rows16<NC>(cs, cq, lane, stats, TN * 32);
}

// feature-set dispatch shared by the GEMM and the patch-staged convolution.  `interior`: every row of the wave block
// and every column of its span is valid; `wave_rows`: rows per wave block (one sample per block is required for the
// row vector).  Returns false when the generic epilogue has to run.
template <int TM, int TN, class MOF>
UDT_DEVINL bool epilogue8_fast_dispatch(const GemmParams& p, f32x16 (&acc)[TM][TN], MOF mof, int nw, int batch, int lane,
                                        char* wlds, bool interior, float* stats) {
  constexpr int PUB = UDT_GEMM_OUT_F32 | UDT_GEMM_GEGLU | UDT_GEMM_RELU | UDT_GEMM_SILU_OUT | (1 << 30) | (1 << 24) |
                      (1 << 25) | (1 << 26);
  if (!interior || UDT_DBG(p.flags, 22)) return false;
  const int f = p.flags & PUB;
  if constexpr (TN == 2) {
    if (f == UDT_GEMM_GEGLU && p.bias && !p.res && !p.rowvec) {
      epilogue8_geglu_fast<TM>(p, acc, mof, nw, batch, lane, wlds);
      return true;
    }
  }
  if (f != 0) return false;
  if (p.rowvec && (p.rows_per_batch % (TM * 32)) != 0) return false;
  const int code = (p.bias ? 1 : 0) | (p.rowvec ? 2 : 0) | (p.res ? 4 : 0);
  if constexpr (TN == 2) {
    switch (code) {
      case 0: epilogue8_rows_fast<TM, false, false, false>(p, acc, mof, nw, batch, lane, wlds, stats); return true;
      case 1: epilogue8_rows_fast<TM, true, false, false>(p, acc, mof, nw, batch, lane, wlds, stats); return true;
      case 3: epilogue8_rows_fast<TM, true, true, false>(p, acc, mof, nw, batch, lane, wlds, stats); return true;
      case 5: epilogue8_rows_fast<TM, true, false, true>(p, acc, mof, nw, batch, lane, wlds, stats); return true;
      default: return false;
    }
  } else {
    switch (code) {
      case 0: epilogue8_rows16_fast<TN, false, false, false>(p, acc, mof, nw, batch, lane, wlds, stats); return true;
      case 1: epilogue8_rows16_fast<TN, true, false, false>(p, acc, mof, nw, batch, lane, wlds, stats); return true;
      case 3: epilogue8_rows16_fast<TN, true, true, false>(p, acc, mof, nw, batch, lane, wlds, stats); return true;
      case 5: epilogue8_rows16_fast<TN, true, false, true>(p, acc, mof, nw, batch, lane, wlds, stats); return true;
      default: return false;
    }
  }
}

UDT_DEVINL void raw_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// WGM x WGN waves, each TM x TN MFMA tiles of 32x32
// STATS: the row-coalesced epilogues also emit the output's column statistics (udt_gemm_desc.colstats) — a separate
// kernel, so that the plain one keeps its instruction stream and register allocation
// (round 5: the e4m3 instance of this kernel — round 2's first-generation fp8 path, unit block scales + a per-tensor activation
// scale — is gone: config #5 runs on the lean family's MX8 instances, lean.h)
template <int WGM, int WGN, int TM, int TN, bool CONV, bool TRANS, bool STATS>
__global__ void __launch_bounds__(NTHREADS) gemm8_kernel(const Params pp) {
  constexpr int EB = 2;                            // bytes per operand element
  static_assert(WGM * WGN == 8, "8 waves per workgroup");
  constexpr int BM = WGM * TM * 32;
  constexpr int BN = WGN * TN * 32;
  static_assert(BM == 256, "A staging assumes 32 pieces");
  constexpr int A_BYTES = BM * ROW_BYTES;
  constexpr int B_BYTES = BN * ROW_BYTES;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int A_INSTR = 4;                       // 32 pieces / 8 waves
  constexpr int B_PIECES = BN / 8;                 // 16 or 20
  constexpr int B_INSTR = (B_PIECES + 7) / 8;      // 2 or 3 per wave (padded with duplicate pieces -> uniform count)
  constexpr int LPT = A_INSTR + B_INSTR;           // loads per K-tile and wave
  constexpr bool ROWS_EPI = (TN == 2) && (TM == 2) && !TRANS;    // row-coalesced epilogue through LDS (256x128 tile)
  constexpr bool ROWS16_EPI = (TM == 1) && !TRANS;               // two 16-row passes (256x160 tile)
  // 256x128: 8 KiB per wave = stage 2 (48 KiB) + 16 KiB above the ring; 256x160: 16 rows x 336 B per wave, in stage 2
  constexpr int EPI_WAVE_BYTES = ROWS16_EPI ? 16 * (TN * 64 + 16) + TN * 128 : TM * 32 * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const GemmParams& p = pp.g;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int hi = lane >> 5;
  const int l3 = lane >> 3;
  const int pslot = lane & 7;
  const int wm = wave / WGN;
  const int wn = wave - wm * WGN;
  const int row0 = wm * TM * 32;
  const int col0 = wn * TN * 32;
  const int swz = (l31 >> 1) & 7;
  const int a_frag_row = (row0 + l31) * ROW_BYTES;
  const int b_frag_row = (col0 + l31) * ROW_BYTES;

  const int g = range_index(blockIdx.x, p.G);
  long long it = (long long)g * p.iters_per_wg;
  long long it_end = it + p.iters_per_wg;
  if (it_end > p.total_iters) it_end = p.total_iters;
  if (it >= it_end) return;

  const int Ctot = p.C1 + p.C2;
  const int Hv = p.Hin << p.ups;
  const int Wv = p.Win << p.ups;

  // ---- per-lane staging state of the current segment ----------------------------------------------------
  // plain GEMM: every load is (descriptor, fixed per-lane voffset, scalar k-offset) -> no per-K-tile vector math.
  // conv gather: the per-lane source pixel changes with the tap, so A keeps 64-bit pointers + a zero page.
  __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.a), 0, pp.a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w), 0, pp.w_bytes, 0x00020000);
  unsigned a_voff[A_INSTR];
  unsigned w_voff[B_INSTR];
  int w_piece[B_INSTR];
  int a_koff[A_INSTR];
  int a_iy0[A_INSTR], a_ix0[A_INSTR], a_pixb[A_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int row = (wave * A_INSTR + i) * 8 + l3;
    a_koff[i] = (pslot ^ ((row >> 1) & 7)) * 8;
  }
#pragma unroll
  for (int i = 0; i < B_INSTR; ++i) {
    int idx = wave + 8 * i;
    while (idx >= B_PIECES) idx -= 8;                      // duplicate piece (re-writes identical bytes)
    w_piece[i] = idx;
  }

  auto prepare = [&](int batch, int m0, int n0) {
    if constexpr (!CONV) {
      rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(p.a) + (long long)batch * p.sA * EB), 0, pp.a_bytes, 0x00020000);
      rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(p.w) + (long long)batch * p.sW * EB), 0, pp.w_bytes, 0x00020000);
    }
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
      const int m = m0 + (wave * A_INSTR + i) * 8 + l3;
      if constexpr (CONV) {
        const int hw = p.Hout * p.Wout;
        const int b = m / hw;
        const int rem = m - b * hw;
        const int oy = rem / p.Wout;
        const int ox = rem - oy * p.Wout;
        a_pixb[i] = b * p.Hin * p.Win;
        a_iy0[i] = (m < p.M) ? (oy * p.stride - p.pad_t) : -100000;
        a_ix0[i] = ox * p.stride - p.pad_l;
        a_voff[i] = 0;
      } else {
        a_voff[i] = (m < p.M) ? (unsigned)((long long)m * p.lda * EB + a_koff[i] * 2) : OOB;
        a_iy0[i] = a_ix0[i] = a_pixb[i] = 0;
      }
    }
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
      const int row = w_piece[i] * 8 + l3;
      const int koff = (pslot ^ ((row >> 1) & 7)) * 8;
      const int n = n0 + row;
      w_voff[i] = (n < p.N) ? (unsigned)((long long)n * p.ldw * EB + koff * 2) : OOB;
    }
  };

  auto stage = [&](int st, int kt) {
    char* abuf = smem + st * STAGE_BYTES;
    char* bbuf = abuf + A_BYTES;
    const int k0 = kt * BK;
    if constexpr (CONV) {
      const int tap = k0 / Ctot;
      const int c0 = k0 - tap * Ctot;
      const int ky = tap / p.ksz;
      const int kx = tap - ky * p.ksz;
      const bool second = c0 >= p.C1;
      const uint16_t* src = second ? p.a2 : p.a;
      const int cs = second ? p.C2 : p.C1;
      const int cc = second ? (c0 - p.C1) : c0;
#pragma unroll
      for (int i = 0; i < A_INSTR; ++i) {
        const int iy = a_iy0[i] + ky;
        const int ix = a_ix0[i] + kx;
        const bool ok = ((unsigned)iy < (unsigned)Hv) && ((unsigned)ix < (unsigned)Wv);
        const long long pix = (long long)a_pixb[i] + (long long)(iy >> p.ups) * p.Win + (ix >> p.ups);
        const uint16_t* gp = ok ? (src + pix * cs + cc + a_koff[i]) : p.zero;
        glds16(gp, abuf + (wave * A_INSTR + i) * 1024);
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_INSTR; ++i) buf_lds16(rsrc_a, abuf + (wave * A_INSTR + i) * 1024, a_voff[i], kt * ROW_BYTES);
    }
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) buf_lds16(rsrc_w, bbuf + w_piece[i] * 1024, w_voff[i], kt * ROW_BYTES);
  };

  int tile = (int)(it / p.n_ktiles);
  int kt0 = (int)(it - (long long)tile * p.n_ktiles);
  int batch, m0, n0;
  decode_tile<BM, BN>(p, tile, batch, m0, n0);
  prepare(batch, m0, n0);
  {
    int kt1 = p.n_ktiles;
    if ((long long)(kt1 - kt0) > it_end - it) kt1 = kt0 + (int)(it_end - it);
    stage(0, kt0);
    if (kt0 + 1 < kt1) stage(1, kt0 + 1);
  }

  // after a register epilogue of an interior tile exactly `nst` stores per wave are YOUNGER than the next segment's
  // two prefetched K-tiles: vmcnt retires in order, so leaving them in flight (LPT + nst) lets the store drain
  // overlap the first two K-tiles instead of stalling them
  int boost = 0, nst = 0;
  while (true) {
    int kt1 = p.n_ktiles;
    if ((long long)(kt1 - kt0) > it_end - it) kt1 = kt0 + (int)(it_end - it);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- K loop: tiles kt and kt+1 are in flight on entry of iteration kt -------------------------------
    // (a lead/lag role split between the two waves of a SIMD was measured: no gain over this straight loop)
    int st = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
      if (kt + 1 < kt1) {                          // leaves exactly one K-tile in flight
        if (boost > 0) {
          --boost;
          if (nst == TM * TN * 4) wait_vm<LPT + TM * TN * 4>();
          else wait_vm<LPT + TM * 4>();
        } else {
          wait_vm<LPT>();
        }
      } else {
        wait_vm<0>();
      }
      raw_barrier();                               // tile kt visible to all waves; stage (st+2)%3 no longer being read
      int st2 = st + 2;
      if (st2 >= NSTAGE) st2 -= NSTAGE;
      if (kt + 2 < kt1) stage(st2, kt + 2);
      const char* abuf = smem + st * STAGE_BYTES;
      const char* bbuf = abuf + A_BYTES;
      {
      // all fragment reads of the K-tile are issued ahead of its MFMAs (sched_group_barrier pins that order: left
      // alone, the scheduler keeps three fragment registers and drains lgkmcnt to 0 twice per 16-wide step, exposing
      // the LDS latency eight times per K-tile); the compiler's counted lgkmcnt waits then release the MFMAs as
      // their operands land
      bf16x8_t xf[4][TM], wf[4][TN];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int slot = ((ks * 2 + hi) ^ swz) << 4;
#pragma unroll
        for (int t = 0; t < TM; ++t) xf[ks][t] = lds_read_frag(abuf + a_frag_row + t * 32 * ROW_BYTES + slot);
#pragma unroll
        for (int t = 0; t < TN; ++t) wf[ks][t] = lds_read_frag(bbuf + b_frag_row + t * 32 * ROW_BYTES + slot);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            if constexpr (TRANS)
              acc[tm][tn] = mfma32(xf[ks][tm], wf[ks][tn], acc[tm][tn]);
            else
              acc[tm][tn] = mfma32(wf[ks][tn], xf[ks][tm], acc[tm][tn]);
          }
      }
      // (plain GEMMs: deep-K shapes +3..6 %, same-box; the gathered convolutions and conv3p measured 1..8 % slower
      //  with the read burst, so they keep the scheduler's interleaving)
      if constexpr (!CONV) {
        __builtin_amdgcn_sched_group_barrier(0x100, 4 * (TM + TN), 0);    // DS reads
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);      // MFMAs
      }
      }
      st = st + 1;
      if (st >= NSTAGE) st = 0;
    }

    // (bits 28 / 27 of flags: measurement modes of udt_debug_set — no slab exchange / no epilogue; wrong results)
    const bool noxchg = UDT_DBG(p.flags, 28);
    const bool full = noxchg || ((kt0 == 0) && (kt1 == p.n_ktiles));
    const bool publish = !noxchg && (kt0 > 0);            // head segment of a tile another workgroup started
    const int cur_tile = tile, cur_batch = batch, cur_m0 = m0, cur_n0 = n0;
    boost = 0;
    it += kt1 - kt0;
    const bool more = it < it_end;
    if (more) {
      // prefetch the next segment's first two K-tiles before finishing this one
      raw_barrier();                                      // every wave is done reading the ring
      tile = (int)(it / p.n_ktiles);
      kt0 = (int)(it - (long long)tile * p.n_ktiles);
      decode_tile<BM, BN>(p, tile, batch, m0, n0);
      prepare(batch, m0, n0);
      int nk1 = p.n_ktiles;
      if ((long long)(nk1 - kt0) > it_end - it) nk1 = kt0 + (int)(it_end - it);
      stage(0, kt0);
      if (kt0 + 1 < nk1) stage(1, kt0 + 1);
    }

    if (publish) {
      // park the accumulators: slab[g][(tm*TN+tn)*4+q][tid] float4 (coalesced 8 KiB per unit), then raise the flag
      f32x4* slab = reinterpret_cast<f32x4*>(pp.slab_base + (long long)g * (BM * BN));
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v = {acc[tm][tn][q * 4 + 0], acc[tm][tn][q * 4 + 1], acc[tm][tn][q * 4 + 2],
                       acc[tm][tn][q * 4 + 3]};
            store16_sc1(slab + ((tm * TN + tn) * 4 + q) * NTHREADS + tid, v);
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // every storing wave drains its stores
      __syncthreads();
      if (tid == 0) {
        // slab stores were write-through (sc1) and are drained: no L2 write-back fence needed
        __hip_atomic_store(pp.flags + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (more) {
        // the drain above also retired the prefetch; nothing else to do (its data is simply already there)
      }
    } else if (!UDT_DBG(p.flags, 27)) {
      if (!full) {
        // this workgroup owns the start of the tile: collect the partners' slabs
        const long long tile_end = ((long long)cur_tile + 1) * p.n_ktiles;
        const int g_last = (int)((tile_end - 1) / p.iters_per_wg);
        // (a partner that never shows up — a launch that was not fully resident — must not be summed: the finisher
        //  sets the workspace's err word, leaves the flags alone and poisons its tile; udt_check_async_error reports it)
        int* const bcast = reinterpret_cast<int*>(smem + 2 * STAGE_BYTES);   // ring stage 2 is idle here
        if (!more) __syncthreads();        // (with `more` the barrier ahead of the prefetch already closed the ring)
        if (tid == 0) {
          int ok = 1;
          for (int pg = g + 1; pg <= g_last && ok; ++pg) {
            int spins = 0;
            while (__hip_atomic_load(pp.flags + pg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
              __builtin_amdgcn_s_sleep(8);
              if (++spins > SPIN_LIMIT) {
                __hip_atomic_store(pp.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
              }
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          *bcast = ok;
        }
        __syncthreads();
        const bool partners_ok = *bcast != 0;
        if (partners_ok) {
          for (int pg = g + 1; pg <= g_last; ++pg) {
            const f32x4* slab = reinterpret_cast<const f32x4*>(pp.slab_base + (long long)pg * (BM * BN));
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
              for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const f32x4 v = slab[((tm * TN + tn) * 4 + q) * NTHREADS + tid];
#pragma unroll
                  for (int r = 0; r < 4; ++r) acc[tm][tn][q * 4 + r] += v[r];
                }
          }
        } else {
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[tm][tn][r] = __builtin_nanf("");
        }
        __syncthreads();
        if (tid == 0 && partners_ok)
          for (int pg = g + 1; pg <= g_last; ++pg)
            __hip_atomic_store(pp.flags + pg, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if constexpr (ROWS16_EPI) {
        if (!(p.flags & (UDT_GEMM_OUT_F32 | UDT_GEMM_GEGLU | (1 << 30)))) {
          if (!more) raw_barrier();
          const int mwv = cur_m0 + row0;
          auto mof = [&](int row) -> long long { return (mwv + row < p.M) ? (long long)(mwv + row) : -1LL; };
          char* wl = smem + 2 * STAGE_BYTES + wave * EPI_WAVE_BYTES;
          const bool interior = (cur_m0 + BM <= p.M) && (cur_n0 + BN <= p.N);
          float* stats = nullptr;
          if constexpr (STATS) stats = p.colstats + ((long long)(mwv / (TM * 32)) * p.N + cur_n0 + col0) * 2;
          if (!epilogue8_fast_dispatch<TM, TN>(p, acc, mof, cur_n0 + col0, cur_batch, lane, wl, interior, stats))
            epilogue8_rows16<TN>(p, acc, mof, cur_n0 + col0, cur_batch, lane, wl, stats);
        } else {
          epilogue8<TM, TN, TRANS>(p, acc, cur_m0, cur_n0, cur_batch, row0, col0, lane);
        }
      } else if constexpr (ROWS_EPI) {
        if (!(p.flags & (UDT_GEMM_OUT_F32 | (1 << 30)))) {
          if (!more) raw_barrier();     // (with `more` the barrier ahead of the prefetch already closed the ring)
          // ring stage 2 (+ the 16 KiB above the ring) is idle here: the next segment's prefetch went to stages 0, 1
          const int mwv = cur_m0 + row0;
          auto mof = [&](int row) -> long long { return (mwv + row < p.M) ? (long long)(mwv + row) : -1LL; };
          char* wl = smem + 2 * STAGE_BYTES + wave * EPI_WAVE_BYTES;
          const bool interior = (cur_m0 + BM <= p.M) && (cur_n0 + BN <= p.N);
          float* stats = nullptr;
          if constexpr (STATS) stats = p.colstats + ((long long)(mwv / (TM * 32)) * p.N + cur_n0 + col0) * 2;
          if (!epilogue8_fast_dispatch<TM, TN>(p, acc, mof, cur_n0 + col0, cur_batch, lane, wl, interior, stats))
            epilogue8_rows<TM>(p, acc, mof, cur_n0 + col0, cur_batch, lane, wl, stats);
        } else {
          epilogue8<TM, TN, TRANS>(p, acc, cur_m0, cur_n0, cur_batch, row0, col0, lane);
        }
      } else {
        epilogue8<TM, TN, TRANS>(p, acc, cur_m0, cur_n0, cur_batch, row0, col0, lane);
      }
      if (!ROWS_EPI && !ROWS16_EPI && full && cur_m0 + BM <= p.M && cur_n0 + BN <= p.N) {
        boost = 2;
        nst = (p.flags & UDT_GEMM_GEGLU) ? TM * 4 : TM * TN * 4;
      }
    }
    if (!more) break;
  }
}

}  // namespace g8
