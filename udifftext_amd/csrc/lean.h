// lean.h — whole-tile MFMA GEMM with CO-RESIDENT workgroups (included by gemm.hip, round 3).
//
// Why a second GEMM family.  The 8-wave kernels of gemm8.h own a CU (160 KiB of LDS, 256 VGPRs + 60..150 spilled
// dwords: one template carries stream-K finishing, six epilogue forms, statistics, fp8 and the gathered convolution),
// so nothing overlaps a workgroup's prologue (HBM latency), its epilogue (LDS transposition, residual loads, GELU) or
// its barrier skew: profiles/r02_gemm_shapes_tile_order.txt — 32768x320x320 takes 29 us against a 7 us HBM floor, the
// class runs at 16 % of the MFMA peak.  This family is the opposite trade:
//   * 4 waves, a 128x128 (or 128x160) tile, a TWO-stage ring: 64 (80) KiB of LDS -> TWO workgroups per CU, which fall
//     into anti-phase on their own (one waits for its LDS-DMA / runs its epilogue while the other owns the matrix
//     pipe), and co-reside with the workgroups of OTHER launch streams (three batches in flight);
//   * one tile (or one K slice of a tile) per workgroup, no waiting between workgroups anywhere: split-K slices park
//     fp32 slabs (write-through stores) and draw a ticket; the LAST arriver sums all slabs in slice order
//     (deterministic) and runs the epilogue — no residency requirement, no spin, no time-out path;
//   * one epilogue: the accumulators go through LDS as fp32 rows, so bias / per-sample row vector / residual are
//     added in the ROW layout (a lane owns 8 consecutive columns: their bias lives in 8 registers) before the single
//     bf16 rounding, and every store is 16 bytes per lane with whole 128-byte lines per row;
//   * separate kernels per feature set (plain / GEGLU, LayerNorm-folded) instead of run-time flags: ~150 VGPRs, no
//     scratch.
// LayerNorm prologue (udt_ln_gemm_fwd, reference attention.py:310-339 `attn1(norm1(x))`, `ff(norm3(x))`): LayerNorm is
// affine per row, so  LN(x) W^T = rstd_m * (x W'^T - mean_m * s) + c  with W' = gamma o W (folded when the weights are
// packed), s_n = sum_k W'_nk, c_n = sum_k beta_k W_nk + bias_n.  The kernel multiplies the RAW rows and accumulates
// every row's sum and sum of squares from the A fragments it reads anyway (v_dot2_f32_bf16: 2 VALU ops per 8
// elements, beside the MFMAs); the epilogue applies rstd / mean.  The normalised activation never exists in memory.
#pragma once

namespace lg {

using g8::wait_vm;
// Workgroup barrier of the lean kernels.  hipcc sinks the MFMAs that consume a K-tile's LAST fragment reads below a bare
// s_barrier (register-only instructions are not ordered by the memory clobber), so those ds_reads would still be in
// flight when, right behind the barrier, another wave's LDS-DMA starts refilling the very stage they read — a rare,
// timing-dependent corruption that showed up as run-to-run differences with two launch streams sharing the CUs.  Every
// wave therefore drains its LDS reads (lgkmcnt only — never vmcnt: the prefetched K-tiles stay in flight) before it
// arrives (cdna_hip_programming.md §5 "Pipelining across barriers").
UDT_DEVINL void raw_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
using g8::buf_lds16;
using g8::OOB;

typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
UDT_DEVINL float dot2_bf16(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, a), __builtin_bit_cast(bf2_t, b), c, false);
}

// Column statistics of a wave's output rows for the consumer's GroupNorm (udt_gemm_desc.colstats; the 8-wave kernels' STATS
// epilogues emit the same [slot][N][2] records): every lane has summed its 8 columns over the rows it stored; the lanes
// that share a column group (rl = 0 .. RPI-1) are combined through the wave's own staging block (a wave's LDS operations
// execute in order, no barrier) in a fixed order, and the rl = 0 lanes write 8 x (sum, sum of squares).
template <int CPR, int RPI>
UDT_DEVINL void wave_colstats(char* wl, int lane, const float (&cs)[8], const float (&cq)[8], float* dst, bool col_ok) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    f32x4 a = {cs[4 * h + 0], cs[4 * h + 1], cs[4 * h + 2], cs[4 * h + 3]};
    f32x4 b = {cq[4 * h + 0], cq[4 * h + 1], cq[4 * h + 2], cq[4 * h + 3]};
    *reinterpret_cast<f32x4*>(wl + lane * 64 + h * 16) = a;
    *reinterpret_cast<f32x4*>(wl + lane * 64 + 32 + h * 16) = b;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane < CPR && col_ok) {
    float ts[8], tq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ts[j] = tq[j] = 0.f;
#pragma unroll
    for (int r = 0; r < RPI; ++r) {
      const char* src = wl + (r * CPR + lane) * 64;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src + h * 16);
        const f32x4 b = *reinterpret_cast<const f32x4*>(src + 32 + h * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) { ts[4 * h + e] += a[e]; tq[4 * h + e] += b[e]; }
      }
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      f32x4 v = {ts[2 * h], tq[2 * h], ts[2 * h + 1], tq[2 * h + 1]};
      *reinterpret_cast<f32x4*>(dst + h * 4) = v;
    }
  }
}

// NW waves as WGM x WGN, each TM x TN MFMA tiles of 32x32; NST ring stages; GEGLU: weight rows packed [32 x | 32 gate]
// per 64-column wave block (TN == 2); LN: LayerNorm folded into the weights, row statistics from the A fragments
// FP8 (udt_gemm_desc UDT_GEMM_MX8, BASELINE config #5): A and W are OCP e4m3 bytes — the LDS image is the same 128-byte rows, now
// 128 K-elements per tile = two v_mfma_scale_f32_32x32x64_f8f6f4 steps — A carries MX block scales (one E8M0 byte per 32
// K-elements of a row, fetched per K-tile as ONE dword per row straight into a register), W a per-output-channel fp32 scale that
// multiplies the accumulators in the epilogue; twice the FLOPs per staged byte, per fragment read and per matrix-pipe cycle.
// With LN the row statistics cannot come from the e4m3 fragments (no packed dot on bytes): the PRODUCER of A emitted partial row
// sums (rowstat_in), summed here in a fixed order.
// EMIT: the epilogue also writes its result as an MX8 activation for the next GEMM (q8_out / q8_scale; GEGLU: `out` may then be
// null — the hidden activation exists as e4m3 only) and, for a LayerNorm-folded consumer, the partial row statistics (rowstat_out).
template <int NW, int WGM, int WGN, int TM, int TN, int NST, bool GEGLU, bool LN, int TMB = TM, bool STATS = false, bool FP8 = false,
          bool EMIT = false>
__global__ void __launch_bounds__(NW * 64, 2) lgemm_kernel(const LParams p) {
  static_assert(!STATS || (!GEGLU && !LN), "statistics-emitting epilogue: plain linears / 1x1 convolutions");
  static_assert(!FP8 || NST == 2, "the fp8 loop is written for the two-stage ring");
  static_assert(!EMIT || (TN * 4) % 4 == 0, "a 32-column block = 4 lanes of the row layout");
  constexpr int EB = FP8 ? 1 : 2;                       // bytes per operand element
  static_assert(TM % TMB == 0, "epilogue passes of TMB row tiles");
  static_assert(WGM * WGN == NW, "wave grid");
  static_assert(!GEGLU || TN == 2, "GEGLU pairs the two 32-column tiles of a wave");
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int A_PIECES = BM / 8, B_PIECES = BN / 8;
  constexpr int A_INSTR = (A_PIECES + NW - 1) / NW, B_INSTR = (B_PIECES + NW - 1) / NW;
  constexpr int LPT = A_INSTR + B_INSTR;
  constexpr int WROWS = TM * 32, WCOLS = TN * 32;
  constexpr int EROW = WCOLS * 4;                       // bytes of one fp32 row of the wave block
  constexpr int EPI_WAVE = TMB * 32 * EROW;             // fp32 staging rows of ONE epilogue pass (TMB of the wave's TM row tiles);
                                                        // LN: + [BM][mean, rstd] behind the NW wave blocks
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int hi = lane >> 5;
  const int l3 = lane >> 3;
  const int pslot = lane & 7;
  const int wm = wave / WGN;
  const int wn = wave - wm * WGN;
  const int row0 = wm * WROWS;
  const int col0 = wn * WCOLS;
  const int swz = (l31 >> 1) & 7;

  const int unit = range_index(blockIdx.x, p.G);
  if (unit >= p.tiles * p.splitk) return;
  const int tile = unit / p.splitk;
  const int slice = unit - tile * p.splitk;
  int m0, n0;
  {
    // blocked tile order (decode_tile of gemm.hip): N-tiles in blocks of n_block, N-fastest inside a block
    const int per_block = p.tiles_m * p.n_block;
    const int blk = tile / per_block;
    const int r = tile - blk * per_block;
    int nbw = p.tiles_n - blk * p.n_block;
    if (nbw > p.n_block) nbw = p.n_block;
    const int tmi = r / nbw;
    m0 = tmi * BM;
    n0 = (blk * p.n_block + (r - tmi * nbw)) * BN;
  }
  const int kt0 = slice * p.kt_per;
  int kt1 = kt0 + p.kt_per;
  if (kt1 > p.nkt) kt1 = p.nkt;
  // (measured and rejected, profiles/r05_krot_rejected.txt: rotating every tile's K range — tile (tm, tn) starts at K-tile
  //  ((tm + tn) mod 8) * nk / 8 and wraps — so that the eight co-resident workgroups that stream the same A rows / weight rows ask for
  //  DIFFERENT K-tiles of it at any moment: 8192x640x2560 37 -> 54 us, 2048x1280x5120 42 -> 61 us, nothing gained anywhere.  The
  //  lock-step is what makes the sharing work: eight simultaneous requests for one line are one fetch; staggered, the XCD's L2 has to
  //  hold eight K-tiles of every shared operand instead of one)
  const int nk = kt1 - kt0;
  auto kreal = [&](int i) { return kt0 + i; };           // i-th K-tile this workgroup processes
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.a), 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w), 0, p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_a2 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.a2 ? p.a2 : p.a), 0, p.a2 ? p.a2_bytes : 0u, 0x00020000);
  unsigned a_voff[A_INSTR], a2_voff[A_INSTR], w_voff[B_INSTR];
  int a_piece[A_INSTR], w_piece[B_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    int idx = wave + NW * i;
    while (idx >= A_PIECES) idx -= NW;                   // duplicate piece (re-writes identical bytes): uniform load count
    a_piece[i] = idx;
    const int row = idx * 8 + l3;
    const int koff = (pslot ^ ((row >> 1) & 7)) * 8;
    const int m = m0 + row;
    a_voff[i] = (m < p.M) ? (unsigned)((long long)m * p.lda * EB + koff * 2) : OOB;
    a2_voff[i] = (m < p.M) ? (unsigned)(((long long)m * p.lda2 + koff) * 2) : OOB;
  }
#pragma unroll
  for (int i = 0; i < B_INSTR; ++i) {
    int idx = wave + NW * i;
    while (idx >= B_PIECES) idx -= NW;
    w_piece[i] = idx;
    const int row = idx * 8 + l3;
    const int koff = (pslot ^ ((row >> 1) & 7)) * 8;
    const int n = n0 + row;
    w_voff[i] = (n < p.N) ? (unsigned)((long long)n * p.ldw * EB + koff * 2) : OOB;
  }
  auto stage = [&](int st, int kt) {
    char* abuf = smem + st * STAGE_BYTES;
    char* bbuf = abuf + A_BYTES;
    if (kt < p.kt_split) {                               // (wave-uniform: a K tile lies in exactly one source)
#pragma unroll
      for (int i = 0; i < A_INSTR; ++i) buf_lds16(rsrc_a, abuf + a_piece[i] * 1024, a_voff[i], kt * ROW_BYTES);
    } else {
#pragma unroll
      for (int i = 0; i < A_INSTR; ++i) buf_lds16(rsrc_a2, abuf + a_piece[i] * 1024, a2_voff[i], (kt - p.kt_split) * ROW_BYTES);
    }
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) buf_lds16(rsrc_w, bbuf + w_piece[i] * 1024, w_voff[i], kt * ROW_BYTES);
  };

  // (measured and rejected, profiles/r03_trace_prefetch_wave.txt: touching the operand lines of later K-tiles to pull them
  //  into L2 ahead of the two-stage ring — from the MFMA waves (vmcnt retires in order: the touch gates the next LDS-DMA
  //  wait) or from a dedicated fifth wave — made the UNet's GEMMs 10-60 % SLOWER: a touch is 64 separate line requests)
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) stage(s, kreal(s));
  // FP8: the block scales of this lane's rows, one dword per row and K-tile (the 4 blocks of the 128-element tile), fetched one tile ahead
  const uint32_t* asp[TM];
  uint32_t asc_nxt[TM];
  if constexpr (FP8) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      int m = m0 + row0 + tm * 32 + l31;
      if (m >= p.M) m = p.M - 1;                         // (rows past M stage zeros: any scale will do)
      asp[tm] = p.a_scale + m;
      asc_nxt[tm] = (nk > 0) ? asp[tm][(long long)kreal(0) * p.M] : 0x7f7f7f7fu;
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float rs[TM], rq[TM];                                 // LN: this lane's half-K share of sum x, sum x^2 of its rows
#pragma unroll
  for (int i = 0; i < TM; ++i) rs[i] = rq[i] = 0.f;
  if constexpr (LN && FP8) {
    // row statistics from the producer's partial sums: the two half-waves take alternate parts (xor32_sum below joins them)
    if (WGN == 1 || wn == 0) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        int m = m0 + row0 + tm * 32 + l31;
        if (m >= p.M) m = p.M - 1;
        const f32x2* sp = reinterpret_cast<const f32x2*>(p.rowstat_in) + m;
        for (int pp = hi; pp < p.rowstat_in_parts; pp += 2) {
          const f32x2 v = sp[(long long)pp * p.M];
          rs[tm] += v[0];
          rq[tm] += v[1];
        }
      }
    }
  }

  const int a_frag_row = (row0 + l31) * ROW_BYTES;
  const int b_frag_row = (col0 + l31) * ROW_BYTES;
  int st = 0;
  for (int kt = 0; kt < nk; ++kt) {   // (kt counts this workgroup's steps; kreal(kt) is the K-tile it processes)
    if (NST > 2 && kt + NST - 2 < nk) wait_vm<LPT*(NST > 2 ? NST - 2 : 0)>();
    else wait_vm<0>();
    raw_barrier();                    // K-tile kt visible to all waves; the stage read in the previous iteration is free
    int asc[TM];
    if constexpr (FP8) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) asc[tm] = (int)(asc_nxt[tm] >> (8 * hi));   // byte 2 ks: block 2 ks + hi of the tile
    }
    if (kt + NST - 1 < nk) {
      int s2 = st + NST - 1;
      if (s2 >= NST) s2 -= NST;
      const int kn = kreal(kt + NST - 1);
      stage(s2, kn);
      if constexpr (FP8) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) asc_nxt[tm] = asp[tm][(long long)kn * p.M];
      }
    }
    const char* abuf = smem + st * STAGE_BYTES;
    const char* bbuf = abuf + A_BYTES;
    if constexpr (FP8) {
      // k-step ks = 64 K-elements = the scale blocks 2 ks and 2 ks + 1: a lane feeds 16 bytes of each (common.h mfma32_mx8)
      i32x8_t xf[2][TM], wf[2][TN];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int slot0 = ((ks * 4 + hi) ^ swz) << 4, slot1 = ((ks * 4 + 2 + hi) ^ swz) << 4;
#pragma unroll
        for (int t = 0; t < TM; ++t)
          xf[ks][t] = lds_read_frag32(abuf + a_frag_row + t * 32 * ROW_BYTES + slot0, abuf + a_frag_row + t * 32 * ROW_BYTES + slot1);
#pragma unroll
        for (int t = 0; t < TN; ++t)
          wf[ks][t] = lds_read_frag32(bbuf + b_frag_row + t * 32 * ROW_BYTES + slot0, bbuf + b_frag_row + t * 32 * ROW_BYTES + slot1);
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32_mx8<0>(wf[0][tn], xf[0][tm], acc[tm][tn], asc[tm]);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32_mx8<2>(wf[1][tn], xf[1][tm], acc[tm][tn], asc[tm]);
      if constexpr (TM * TN <= 5) {
        __builtin_amdgcn_sched_group_barrier(0x100, 4 * (TM + TN), 0);    // all fragment reads of the K-tile ...
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * TM * TN, 0);      // ... ahead of its MFMAs
      }
    } else {
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
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32(wf[ks][tn], xf[ks][tm], acc[tm][tn]);
      if constexpr (LN) {
        if (wn == 0) {
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) {
            const u32x4 v = __builtin_bit_cast(u32x4, xf[ks][tm]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              rs[tm] = dot2_bf16(v[j], 0x3f803f80u, rs[tm]);      // (1.0, 1.0)
              rq[tm] = dot2_bf16(v[j], v[j], rq[tm]);
            }
          }
        }
      }
    }
    if constexpr (TM * TN <= 5) {
      __builtin_amdgcn_sched_group_barrier(0x100, 4 * (TM + TN), 0);    // all fragment reads of the K-tile ...
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);      // ... ahead of its MFMAs (gemm8.h)
    }                                                                   // (128 x 64 wave tiles: the scheduler's own interleave)
    }
    st = st + 1;
    if (st >= NST) st = 0;
  }
  raw_barrier();                      // every wave is done with the ring: it becomes the epilogue's staging space

  // ---- split-K: park the slice, draw a ticket; the last arriver sums all slices in order ----------------------------
  if (p.splitk > 1) {
    f32x4* slab = reinterpret_cast<f32x4*>(p.slabs + (long long)unit * (BM * BN));
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v = {acc[tm][tn][q * 4 + 0], acc[tm][tn][q * 4 + 1], acc[tm][tn][q * 4 + 2], acc[tm][tn][q * 4 + 3]};
          store16_sc1(slab + ((tm * TN + tn) * 4 + q) * (NW * 64) + tid, v);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores
    __syncthreads();
    int* const bcast = reinterpret_cast<int*>(smem);
    if (tid == 0) {
      const int t = __hip_atomic_fetch_add(p.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = (t == p.splitk - 1) ? 1 : 0;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(p.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next launch
      }
      *bcast = last;
    }
    __syncthreads();
    const bool last = *bcast != 0;
    __syncthreads();                                      // (bcast lives in the staging space)
    if (!last) return;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int s = 0; s < p.splitk; ++s) {
      const f32x4* sl = reinterpret_cast<const f32x4*>(p.slabs + ((long long)tile * p.splitk + s) * (BM * BN));
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = sl[((tm * TN + tn) * 4 + q) * (NW * 64) + tid];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[tm][tn][q * 4 + r] += v[r];
          }
    }
  }

  // LN row statistics: the two half-waves hold the two halves of K.  One wave column (WGN == 1): every wave owns the
  // statistics of its rows in registers (row r of block tm in lanes r and r + 32) and the row-layout pass fetches them
  // with a lane shuffle; otherwise the wn == 0 wave of each wave row publishes [BM][mean, rstd] behind the wave blocks.
  float ln_mean[TM], ln_rstd[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) ln_mean[tm] = 0.f, ln_rstd[tm] = 1.f;
  if constexpr (LN) {
    float* const rst = reinterpret_cast<float*>(smem + NW * EPI_WAVE);
    if (WGN == 1 || wn == 0) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const float s = xor32_sum(rs[tm]), q = xor32_sum(rq[tm]);
        const float mean = s / (float)p.K;
        const float var = fmaxf(q / (float)p.K - mean * mean, 0.f);
        ln_mean[tm] = mean;
        ln_rstd[tm] = __builtin_amdgcn_rsqf(var + p.ln_eps);
        if constexpr (WGN > 1) {
          if (hi == 0) {
            f32x2 o = {ln_mean[tm], ln_rstd[tm]};
            *reinterpret_cast<f32x2*>(rst + (row0 + tm * 32 + l31) * 2) = o;
          }
        }
      }
    }
    if constexpr (WGN > 1) __syncthreads();
  }
  auto row_stats = [&](int row, float& mean, float& rstd) {       // called by all lanes (shuffle)
    if constexpr (WGN == 1) {
      mean = __shfl(ln_mean[0], row & 31);
      rstd = __shfl(ln_rstd[0], row & 31);
#pragma unroll
      for (int tm = 1; tm < TM; ++tm) {
        const float m2 = __shfl(ln_mean[tm], row & 31), r2 = __shfl(ln_rstd[tm], row & 31);
        if ((row >> 5) == tm) mean = m2, rstd = r2;
      }
    } else {
      const f32x2 ms = *reinterpret_cast<const f32x2*>(reinterpret_cast<const float*>(smem + NW * EPI_WAVE) + (row0 + row) * 2);
      mean = ms[0];
      rstd = ms[1];
    }
  };

  if constexpr (GEGLU) {
    // x and its gate sit in the SAME lane and register index of acc[tm][0] / acc[tm][1]: GEGLU is applied in the
    // accumulator layout and only the bf16 result (32 columns = 64 bytes per row) is transposed through LDS
    if constexpr (LN && WGN > 1) {
      if (wn != 0) {
        const float* rst = reinterpret_cast<const float*>(smem + NW * EPI_WAVE);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          const f32x2 ms = *reinterpret_cast<const f32x2*>(rst + (row0 + tm * 32 + l31) * 2);
          ln_mean[tm] = ms[0];
          ln_rstd[tm] = ms[1];
        }
      }
    }
    char* const wb = smem + wave * (WROWS * 64);
#pragma unroll
    for (int q = 0; q < 4; ++q) {                         // (q outer: one column group's constants live at a time)
      const int n = n0 + col0 + q * 8 + hi * 4;
      f32x4 bx = {0.f, 0.f, 0.f, 0.f}, bg = bx, sx = bx, sg = bx;
      f32x4 wx = {1.f, 1.f, 1.f, 1.f}, wg = wx;             // FP8: the weight rows' scales
      if (n < p.N) {
        if constexpr (FP8) {
          wx = *reinterpret_cast<const f32x4*>(p.colscale + n);
          wg = *reinterpret_cast<const f32x4*>(p.colscale + n + 32);
        }
        if (p.bias) {
          bx = *reinterpret_cast<const f32x4*>(p.bias + n);
          bg = *reinterpret_cast<const f32x4*>(p.bias + n + 32);
        }
        if constexpr (LN) {
          sx = *reinterpret_cast<const f32x4*>(p.ln_s + n);
          sg = *reinterpret_cast<const f32x4*>(p.ln_s + n + 32);
        }
      }
      // operands of geglu_scaled: the scales GEGLU_XS / GEGLU_GS ride on the bias and on the row multipliers
#pragma unroll
      for (int r = 0; r < 4; ++r) bx[r] *= GEGLU_XS, bg[r] *= GEGLU_GS;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int row = tm * 32 + l31;
        const float xm = ln_rstd[tm] * p.alpha * GEGLU_XS, gm = ln_rstd[tm] * p.alpha * GEGLU_GS;   // (not LN: rstd = 1, mean = 0)
        const float xa = -ln_mean[tm] * xm, ga = -ln_mean[tm] * gm;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x, g;
          float ax = acc[tm][0][q * 4 + r], ag = acc[tm][1][q * 4 + r];
          if constexpr (FP8) ax *= wx[r], ag *= wg[r];
          if constexpr (LN) {
            x = xm * ax + (xa * sx[r] + bx[r]);
            g = gm * ag + (ga * sg[r] + bg[r]);
          } else {
            x = xm * ax + bx[r];
            g = gm * ag + bg[r];
          }
          o[r] = geglu_scaled(x, g);
        }
        u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
        *reinterpret_cast<u32x2*>(wb + row * 64 + ((q ^ (row & 3)) << 4) + hi * 8) = pk;
      }
    }
    const int ch = lane & 3;
    const int n_out = ((n0 + col0) >> 1) + ch * 8;
    const bool col_ok = (n0 + col0 + ch * 8) < p.N;
#pragma unroll
    for (int i = 0; i < WROWS / 16; ++i) {
      const int row = i * 16 + (lane >> 2);
      const int m = m0 + row0 + row;
      const u32x4 v = *reinterpret_cast<const u32x4*>(wb + row * 64 + ((ch ^ (row & 3)) << 4));
      if constexpr (EMIT) {
        // the 4 lanes of a quad hold one row's 32 output columns = one MX block (from the bf16-rounded values)
        const float f[8] = {bf16_lo(v[0]), bf16_hi(v[0]), bf16_lo(v[1]), bf16_hi(v[1]), bf16_lo(v[2]), bf16_hi(v[2]), bf16_lo(v[3]), bf16_hi(v[3])};
        uint32_t sb;
        const u32x2 q8 = mx8_quant_row8(f, sb);
        if (m < p.M && col_ok) {
          *reinterpret_cast<u32x2*>(p.q8_out + (long long)m * p.ld_q8 + n_out) = q8;
          if (ch == 0) reinterpret_cast<uint8_t*>(p.q8_scale)[((long long)(n_out >> 7) * p.M + m) * 4 + ((n_out >> 5) & 3)] = (uint8_t)sb;
          if (p.out) *reinterpret_cast<u32x4*>(p.out + (long long)m * p.ldo + n_out) = v;
        }
      } else {
        if (m < p.M && col_ok) *reinterpret_cast<u32x4*>(p.out + (long long)m * p.ldo + n_out) = v;
      }
    }
  } else {
    // ---- epilogue: accumulators -> this wave's fp32 rows in LDS (16-byte chunks XOR-swizzled by row & 7), in TM / TMB
    // passes of TMB * 32 rows (the 256 x 256 tile's accumulators exceed the LDS in one piece); a wave only ever touches
    // its own staging block, and its LDS operations execute in order, so the passes need no barrier
    char* const wl = smem + wave * EPI_WAVE;
    constexpr int PROWS = TMB * 32;                      // rows per pass
    constexpr int CPR = TN * 4;                          // 8-column groups per wave row
    constexpr int RPI = 64 / CPR;                        // rows per instruction (8, or 3 with 4 idle lanes)
    constexpr int NIT = (PROWS + RPI - 1) / RPI;
    const int rl = lane / CPR;
    const int c8 = lane - rl * CPR;
    const int n = n0 + col0 + c8 * 8;
    const bool col_ok = (rl < RPI) && (n < p.N);
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0, s0 = b0, s1 = b0;
    if (p.bias && col_ok) {
      b0 = *reinterpret_cast<const f32x4*>(p.bias + n);
      b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
    }
    if constexpr (LN) {
      if (col_ok) {
        s0 = *reinterpret_cast<const f32x4*>(p.ln_s + n);
        s1 = *reinterpret_cast<const f32x4*>(p.ln_s + n + 4);
      }
    }
    f32x4 w0 = {1.f, 1.f, 1.f, 1.f}, w1 = w0;              // FP8: the weight rows' scales
    if constexpr (FP8) {
      if (col_ok) {
        w0 = *reinterpret_cast<const f32x4*>(p.colscale + n);
        w1 = *reinterpret_cast<const f32x4*>(p.colscale + n + 4);
      }
    }
    float cs[8], cq[8];                                  // STATS: this lane's column sums over the rows it stores
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[j] = cq[j] = 0.f;
#pragma unroll
    for (int pass = 0; pass < TM / TMB; ++pass) {
      const int prow0 = pass * PROWS;                    // first wave row of this pass
      // residual rows first (whole lines, all loads in flight together)
      u32x4 rv[NIT];
      if (p.res) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          const int row = i * RPI + rl;
          const int m = m0 + row0 + prow0 + row;
          u32x4 z = {0u, 0u, 0u, 0u};
          rv[i] = z;
          if (col_ok && row < PROWS && m < p.M) rv[i] = *reinterpret_cast<const u32x4*>(p.res + (long long)m * p.ldr + n);
        }
      }
#pragma unroll
      for (int tb = 0; tb < TMB; ++tb) {
        const int row = tb * 32 + l31;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int chunk = (tn * 8 + q * 2 + hi) ^ (row & 7);
            const f32x16& a16 = acc[pass * TMB + tb][tn];
            f32x4 v = {a16[q * 4 + 0], a16[q * 4 + 1], a16[q * 4 + 2], a16[q * 4 + 3]};
            *reinterpret_cast<f32x4*>(wl + row * EROW + chunk * 16) = v;
          }
      }
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int row = i * RPI + rl;
        const int m = m0 + row0 + prow0 + row;
        const bool ok = col_ok && row < PROWS && m < p.M;
        const int rr = row < PROWS ? row : 0;
        const char* rp = wl + rr * EROW;
        const int sw = rr & 7;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(rp + (((2 * c8) ^ sw) << 4));
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(rp + (((2 * c8 + 1) ^ sw) << 4));
        float o[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = v0[j];
          o[4 + j] = v1[j];
        }
        if constexpr (FP8) {
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] *= w0[j], o[4 + j] *= w1[j];
        }
        if constexpr (LN) {
          float mean, rstd;
          row_stats(prow0 + rr, mean, rstd);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            o[j] = rstd * (o[j] - mean * s0[j]);
            o[4 + j] = rstd * (o[4 + j] - mean * s1[j]);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = o[j] * p.alpha + b0[j];
          o[4 + j] = o[4 + j] * p.alpha + b1[j];
        }
        if (p.rowvec && ok) {
          const float* rvp = p.rowvec + (long long)(m / p.rows_per_batch) * p.ldrv + n;
          const f32x4 r0 = *reinterpret_cast<const f32x4*>(rvp);
          const f32x4 r1 = *reinterpret_cast<const f32x4*>(rvp + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            o[j] += r0[j];
            o[4 + j] += r1[j];
          }
        }
        if (p.res) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            o[2 * j] += bf16_lo(rv[i][j]);
            o[2 * j + 1] += bf16_hi(rv[i][j]);
          }
        }
        if constexpr (EMIT) {
          // the result again as an MX8 activation: the 4 lanes of an aligned quad hold one 32-column block of the row (every lane
          // executes the cross-lane steps; N % 32 == 0 keeps a quad's lanes valid together)
          // ... of the values AS ROUNDED FOR `out` (udt_kernels.h q8_out; the GEGLU and row-resident emitting epilogues do the same):
          // the bf16 residual stream and its e4m3 twin / LayerNorm statistics come from ONE rounding, whichever plan served the layer
          {
            const u32x4 rb = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
#pragma unroll
            for (int j = 0; j < 4; ++j) o[2 * j] = bf16_lo(rb[j]), o[2 * j + 1] = bf16_hi(rb[j]);
          }
          uint32_t sb;
          u32x2 q8 = mx8_quant_row8(o, sb);
          if (n >= p.q8_fixed_col) {                      // (the v third of a q|k|v projection: one scale for the whole tensor)
            q8 = e4m3_fixed_row8(o, p.q8_fixed_mul);
            sb = 127u;
          }
          float ps = 0.f, pq = 0.f;
          if (p.rowstat_out) {                            // partial row statistics over this wave's columns (CPR = 8 or 16 lanes)
            static_assert(CPR == 8 || CPR == 16, "row statistics: 64- or 128-column wave blocks");
#pragma unroll
            for (int j = 0; j < 8; ++j) ps += o[j], pq += o[j] * o[j];
            ps = dpp_add<0xB1>(ps); pq = dpp_add<0xB1>(pq);      // lane ^ 1
            ps = dpp_add<0x4E>(ps); pq = dpp_add<0x4E>(pq);      // lane ^ 2
            ps = dpp_add<0x141>(ps); pq = dpp_add<0x141>(pq);    // row_half_mirror: + the other quad of the 8 lanes
            if constexpr (CPR == 16) { ps = dpp_add<0x140>(ps); pq = dpp_add<0x140>(pq); }   // row_mirror: + the other 8
          }
          if (ok) {
            *reinterpret_cast<u32x2*>(p.q8_out + (long long)m * p.ld_q8 + n) = q8;
            if ((c8 & 3) == 0) reinterpret_cast<uint8_t*>(p.q8_scale)[((long long)(n >> 7) * p.M + m) * 4 + ((n >> 5) & 3)] = (uint8_t)sb;
            if (p.rowstat_out && c8 == 0) {
              f32x2 st2 = {ps, pq};
              reinterpret_cast<f32x2*>(p.rowstat_out)[(long long)((n0 + col0) / WCOLS) * p.M + m] = st2;
            }
          }
        }
        if (ok) {
          u32x4 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
          if (!EMIT || p.out) *reinterpret_cast<u32x4*>(p.out + (long long)m * p.ldo + n) = pk;
          if constexpr (STATS) {                         // statistics of the values as stored (bf16-rounded)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float a = bf16_lo(pk[j]), bb = bf16_hi(pk[j]);
              cs[2 * j] += a; cq[2 * j] += a * a;
              cs[2 * j + 1] += bb; cq[2 * j + 1] += bb * bb;
            }
          }
        }
      }
    }
    if constexpr (STATS) {
      const int slot = (m0 + row0) / WROWS;
      wave_colstats<CPR, RPI>(wl, lane, cs, cq, p.colstats + ((long long)slot * p.N + n) * 2, col_ok);
    }
  }
}


// ---- lean 3x3 / stride 1 / pad 1 convolution (LDS-staged patches, co-resident workgroups) -----------------------------
// The patch-staged formulation of conv3p.h (a workgroup stages the input patch of its pixel tile WITH ITS HALO once per
// 64-channel chunk; the nine taps read their A fragments from it at shifted rows, only the weight tile streams per tap)
// in the lean form: 4 waves, 128 output pixels (16 x 8) x 128 output channels, two patch buffers (23 KiB each) + a
// TWO-stage weight ring (16 KiB each) = 78 KiB -> two workgroups per CU, from the same or from different launch streams;
// whole tiles (or, for the few-tile deep layers, chunk slices with the ticket split-K of lgemm_kernel); the fp32-row
// epilogue of lgemm_kernel (bias + time-embedding row vector + residual before the single bf16 rounding, 16-byte stores of
// whole 128-byte lines).  ~190 VGPRs, no scratch, no stream-K residency contract.
// Replaces nn.Conv2d(k=3, pad=1) of ResBlock / ResnetBlock (reference openaimodel.py:183-187,218-231; model.py:128-148).
// s_waitcnt vmcnt(n) for a run-time n in 0 .. MAXN that the unrolled caller makes a constant
template <int MAXN>
UDT_DEVINL void wait_vm_upto(int n) {
  if constexpr (MAXN == 0) wait_vm<0>();
  else {
    if (n >= MAXN) wait_vm<MAXN>();
    else wait_vm_upto<MAXN - 1>(n);
  }
}

// tile geometry: TW x TH output pixels (128 = 16 x 8: two 32-pixel MFMA tiles per wave; 64 = 8 x 8 for the 8 x 8 maps: one),
// UPS: nearest x2 upsampling folded in (reference Upsample.forward, openaimodel.py:99-101 / model.py:64-68): the tile lives
// in the UPSAMPLED map, the staged patch is the low-resolution region under it with its halo, and tap (dy, dx) of output
// pixel (py, px) reads patch row ((py + dy - 1) >> 1) + 1, column ((px + dx - 1) >> 1) + 1; p.H / p.W are the INPUT size
template <int TW, int TH, bool UPS>
struct C3Geo {
  static constexpr int PX = TW * TH;
  static constexpr int TM = PX / 64;                       // 32-pixel MFMA tiles per wave (wave grid 2 x 2)
  static constexpr int PW = (UPS ? TW / 2 : TW) + 2, PH = (UPS ? TH / 2 : TH) + 2;
  static constexpr int PROWS = PW * PH;
  static constexpr int PIECES = (PROWS + 7) / 8;
  static constexpr int PATCH_BYTES = PIECES * 1024;
  static constexpr int W_BYTES = 128 * ROW_BYTES;
  // weight ring: two stages (one tile in flight while one is consumed); THREE where the smaller patch of the 8 x 8 geometry
  // leaves room inside 80 KiB — those are the deep layers whose weights stream from HBM once per launch (M = 512, K = 11520:
  // 29 MB of weights for 15 GFLOP), i.e. the launches that wait on tile arrival
  static constexpr int NRING = (3 * W_BYTES + 2 * PATCH_BYTES <= 80 * 1024) ? 3 : 2;
  static constexpr int RING = NRING * W_BYTES + 2 * PATCH_BYTES;
  static constexpr int STAGING = 4 * TM * 32 * 256;        // fp32 rows of the epilogue
  static constexpr int SMEM = RING > STAGING ? RING : STAGING;
  static_assert(PX == 128 || PX == 64, "128 or 64 output pixels per workgroup");
  static_assert(SMEM <= 80 * 1024, "two workgroups per CU");
};

template <int TW, int TH, bool UPS, bool STATS = false>
__global__ void __launch_bounds__(256, 2) lconv3_kernel(const C3Params p) {
  using Geo = C3Geo<TW, TH, UPS>;
  constexpr int NW = 4, TM = Geo::TM, TN = 2, BN = 128, BMPX = Geo::PX;
  constexpr int C3_PW = Geo::PW, C3_PROWS = Geo::PROWS, C3_PIECES = Geo::PIECES, C3_PATCH_BYTES = Geo::PATCH_BYTES;
  constexpr int C3_W_BYTES = Geo::W_BYTES, C3_TW = TW, C3_TH = TH;
  constexpr int WROWS = TM * 32;                         // pixels per wave
  constexpr int WP = 4;                                  // weight pieces per wave and tap (16 / 4)
  constexpr int PP = (C3_PIECES + NW - 1) / NW;          // patch pieces per wave and chunk (padded with duplicates)
  constexpr int EROW = 64 * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wring = smem;
  constexpr int NR = Geo::NRING;
  char* const patches = smem + NR * C3_W_BYTES;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int hi = lane >> 5;
  const int l3 = lane >> 3;
  const int pslot = lane & 7;
  const int wm = wave >> 1, wn = wave & 1;
  const int row0 = wm * WROWS, col0 = wn * 64;
  const int swz_w = (l31 >> 1) & 7;
  const int w_frag_row = (col0 + l31) * ROW_BYTES;

  const int unit = range_index(blockIdx.x, p.G);
  if (unit >= p.tiles * p.splitk) return;
  const int tile = unit / p.splitk;
  const int slice = unit - tile * p.splitk;
  int tile_m, n0;
  {
    const int per_block = p.tiles_m * p.n_block;
    const int blk = tile / per_block;
    const int r = tile - blk * per_block;
    int nbw = p.tiles_n - blk * p.n_block;
    if (nbw > p.n_block) nbw = p.n_block;
    tile_m = r / nbw;
    n0 = (blk * p.n_block + (r - tile_m * nbw)) * BN;
  }
  const int per_img = p.tiles_x * p.tiles_y;
  const int b = tile_m / per_img;
  const int rt = tile_m - b * per_img;
  const int ty = rt / p.tiles_x;
  const int y0 = ty * C3_TH, x0 = (rt - ty * p.tiles_x) * C3_TW;
  const int c0 = slice * p.ch_per;
  int c1 = c0 + p.ch_per;
  if (c1 > p.chunks) c1 = p.chunks;

  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.a), 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w), 0, p.w_bytes, 0x00020000);
  // XOR swizzle of a patch row's 16-byte slots.  The rows one ds_read_b128 lane group touches are 16 consecutive pixels of
  // one or two image rows; with the patch pitch of TW + 2 rows, (row >> 1) & 7 maps two of them onto the same slot (PMC:
  // a third of the LDS cycles of the first version were bank conflicts).  Keying the swizzle on the row index with the
  // halo pitch REMOVED (row - 2 * patch line: consecutive image rows then differ by the interior width, a multiple of
  // 8) makes the lanes of a group hit 16 different (bank half, slot) pairs again.
  auto patch_swz = [](int prow, int yy) { return ((prow - 2 * yy) >> 1) & 7; };
  unsigned w_voff[WP], p_voff[PP];
  int p_piece[PP];
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    const int row = (wave + NW * i) * 8 + l3;
    const int kw = (pslot ^ ((row >> 1) & 7)) * 8;
    const int n = n0 + row;
    w_voff[i] = (n < p.N) ? (unsigned)(((long long)n * p.ldw + kw) * 2) : OOB;
  }
#pragma unroll
  for (int i = 0; i < PP; ++i) {
    int idx = wave + NW * i;
    while (idx >= C3_PIECES) idx -= NW;
    p_piece[i] = idx;
    const int prow = idx * 8 + l3;
    const int yy = prow / C3_PW;
    const int xx = prow - yy * C3_PW;
    const int gy = (UPS ? (y0 >> 1) : y0) + yy - 1, gx = (UPS ? (x0 >> 1) : x0) + xx - 1;
    const bool ok = (prow < C3_PROWS) && ((unsigned)gy < (unsigned)p.H) && ((unsigned)gx < (unsigned)p.W);
    const int koff = (pslot ^ patch_swz(prow, yy)) * 8;
    p_voff[i] = ok ? (unsigned)(((((long long)b * p.H + gy) * p.W + gx) * p.C + koff) * 2) : OOB;
  }
  int a_prow[TM], a_py[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int ml = row0 + tm * 32 + l31;
    const int py = ml / TW, px = ml - py * TW;
    a_py[tm] = py;
    a_prow[tm] = UPS ? (py | (px << 16)) : (py * C3_PW + px);      // patch row of this lane's pixel at tap (0, 0)
  }
  auto issue_w = [&](int st, int c, int tap) {
    const int soff = (tap * p.C + c * 64) * 2;
    char* wbuf = wring + st * C3_W_BYTES;
#pragma unroll
    for (int i = 0; i < WP; ++i) buf_lds16(rsrc_w, wbuf + (wave + NW * i) * 1024, w_voff[i], soff);
  };
  auto issue_patch = [&](int c) {
    char* pbuf = patches + (c & 1) * C3_PATCH_BYTES;
#pragma unroll
    for (int i = 0; i < PP; ++i) buf_lds16(rsrc_a, pbuf + p_piece[i] * 1024, p_voff[i], c * 128);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  issue_patch(c0);
#pragma unroll
  for (int j = 0; j < NR - 1; ++j) issue_w(j, c0, j);     // (a chunk has nine taps >= NR - 1)
  int st = 0;
  for (int c = c0; c < c1; ++c) {
    const bool nxt = (c + 1 < c1);
    const char* pbuf = patches + (c & 1) * C3_PATCH_BYTES;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // loads younger than this tap's weight tile, in issue order: the NR - 2 tiles after it (fewer at the very end) and,
      // for taps 1 .. NR - 1 of a chunk that has a successor, the next patch (issued at tap 0 behind that tap's tile): they
      // stay in flight, everything older has landed
      {
        const int w_ahead = nxt ? NR - 2 : ((8 - tap) < NR - 2 ? (8 - tap) : NR - 2);
        const bool patch_ahead = nxt && tap >= 1 && tap <= NR - 1;
        wait_vm_upto<(NR - 2) * WP + PP>(w_ahead * WP + (patch_ahead ? PP : 0));   // (folds to one s_waitcnt per tap)
      }
      raw_barrier();
      if (!UDT_DBG(p.dbg, 0)) {
        const int ahead = tap + NR - 1;
        int s2 = st + NR - 1;
        if (s2 >= NR) s2 -= NR;
        if (ahead < 9) issue_w(s2, c, ahead);
        else if (nxt) issue_w(s2, c + 1, ahead - 9);
      }
      if (tap == 0 && nxt && !UDT_DBG(p.dbg, 1)) issue_patch(c + 1);
      const int dy = tap / 3, dx = tap - dy * 3;
      const char* wbuf = wring + st * C3_W_BYTES;
      bf16x8_t fx[4][TM], fw[4][TN];
      if (!UDT_DBG(p.dbg, 3)) {
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          int prow, yy;
          if constexpr (UPS) {
            const int py = a_prow[t] & 0xffff, px = a_prow[t] >> 16;
            yy = ((py + dy - 1) >> 1) + 1;
            prow = yy * C3_PW + (((px + dx - 1) >> 1) + 1);
          } else {
            yy = a_py[t] + dy;
            prow = a_prow[t] + dy * C3_PW + dx;
          }
          const int arow = prow * ROW_BYTES, aswz = patch_swz(prow, yy);
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) fx[ks][t] = lds_read_frag(pbuf + arow + (((ks * 2 + hi) ^ aswz) << 4));
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int slot = ((ks * 2 + hi) ^ swz_w) << 4;
#pragma unroll
          for (int t = 0; t < TN; ++t) fw[ks][t] = lds_read_frag(wbuf + w_frag_row + t * 32 * ROW_BYTES + slot);
        }
      }
      if (UDT_DBG(p.dbg, 2)) {                           // measurement builds: keep the fragment reads, drop the MFMAs
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) asm volatile("" ::"v"(fx[ks][tm]));
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) asm volatile("" ::"v"(fw[ks][tn]));
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32(fw[ks][tn], fx[ks][tm], acc[tm][tn]);
      }
      st = (st + 1 == NR) ? 0 : st + 1;
    }
  }
  raw_barrier();

  if (p.splitk > 1) {
    f32x4* slab = reinterpret_cast<f32x4*>(p.slabs + (long long)unit * (BMPX * BN));
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v = {acc[tm][tn][q * 4 + 0], acc[tm][tn][q * 4 + 1], acc[tm][tn][q * 4 + 2], acc[tm][tn][q * 4 + 3]};
          store16_sc1(slab + ((tm * TN + tn) * 4 + q) * 256 + tid, v);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* const bcast = reinterpret_cast<int*>(smem);
    if (tid == 0) {
      const int t = __hip_atomic_fetch_add(p.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = (t == p.splitk - 1) ? 1 : 0;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(p.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      *bcast = last;
    }
    __syncthreads();
    const bool last = *bcast != 0;
    __syncthreads();
    if (!last) return;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int s = 0; s < p.splitk; ++s) {
      const f32x4* sl = reinterpret_cast<const f32x4*>(p.slabs + ((long long)tile * p.splitk + s) * (BMPX * BN));
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = sl[((tm * TN + tn) * 4 + q) * 256 + tid];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[tm][tn][q * 4 + r] += v[r];
          }
    }
  }

  // ---- epilogue: fp32 rows through LDS (as lgemm_kernel), one pixel = one row ---------------------------------------
  char* const wl = smem + wave * (WROWS * EROW);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int row = tm * 32 + l31;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int chunk = (tn * 8 + q * 2 + hi) ^ (row & 7);
        f32x4 v = {acc[tm][tn][q * 4 + 0], acc[tm][tn][q * 4 + 1], acc[tm][tn][q * 4 + 2], acc[tm][tn][q * 4 + 3]};
        *reinterpret_cast<f32x4*>(wl + row * EROW + chunk * 16) = v;
      }
  }
  const int rl = lane >> 3, c8 = lane & 7;
  const int n = n0 + col0 + c8 * 8;
  const bool col_ok = n < p.N;
  f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
  if (col_ok) {
    if (p.bias) {
      b0 = *reinterpret_cast<const f32x4*>(p.bias + n);
      b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
    }
    if (p.rowvec) {                                      // one image per tile: the time-embedding row is a per-lane constant
      const float* rvp = p.rowvec + (long long)b * p.ldrv + n;
      const f32x4 r0 = *reinterpret_cast<const f32x4*>(rvp), r1 = *reinterpret_cast<const f32x4*>(rvp + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) b0[j] += r0[j], b1[j] += r1[j];
    }
  }
  constexpr int NIT = TM * 4;                            // 8 pixel rows per instruction
  const int Ho = UPS ? 2 * p.H : p.H, Wo = UPS ? 2 * p.W : p.W;
  long long mrow[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int ml = row0 + i * 8 + rl;
    const int py = ml / TW, px = ml - py * TW;
    mrow[i] = ((long long)b * Ho + (y0 + py)) * Wo + (x0 + px);
  }
  u32x4 rv[NIT];
  if (p.res) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      u32x4 z = {0u, 0u, 0u, 0u};
      rv[i] = z;
      if (col_ok) rv[i] = *reinterpret_cast<const u32x4*>(p.res + mrow[i] * p.ldr + n);
    }
  }
  float cs[8], cq[8];                                    // STATS: this lane's column sums over the pixels it stores
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = cq[j] = 0.f;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int row = i * 8 + rl;
    const char* rp = wl + row * EROW;
    const int sw = row & 7;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(rp + (((2 * c8) ^ sw) << 4));
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(rp + (((2 * c8 + 1) ^ sw) << 4));
    float o[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = v0[j] * p.alpha + b0[j];
      o[4 + j] = v1[j] * p.alpha + b1[j];
    }
    if (p.res) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[2 * j] += bf16_lo(rv[i][j]);
        o[2 * j + 1] += bf16_hi(rv[i][j]);
      }
    }
    if (col_ok) {
      u32x4 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
      *reinterpret_cast<u32x4*>(p.out + mrow[i] * p.ldo + n) = pk;
      if constexpr (STATS) {                             // statistics of the values as stored (bf16-rounded)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = bf16_lo(pk[j]), bb = bf16_hi(pk[j]);
          cs[2 * j] += a; cq[2 * j] += a * a;
          cs[2 * j + 1] += bb; cq[2 * j + 1] += bb * bb;
        }
      }
    }
  }
  if constexpr (STATS) {
    // one slot per wave pixel block (WROWS pixels of one image); the two waves that share the pixels cover different channels
    const int slot = tile_m * 2 + wm;
    wave_colstats<8, 8>(wl, lane, cs, cq, p.colstats + ((long long)slot * p.N + n) * 2, col_ok);
  }
}

}  // namespace lg
