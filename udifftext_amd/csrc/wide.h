// wide.h — 3x3 / stride 1 / pad 1 convolution on WIDE wave tiles (included by gemm.hip, round 4).
//
// Why a third tiling.  The MFMA loops of lean.h are bound by the CU's LDS port, not by the matrix pipe: a wave that owns
// 64 x 64 outputs reads (2 + 2) operand fragments of 1 KiB per 4 MFMAs, and the workgroup's weight tile arrives through
// the same port — 1.0 KiB of fragment reads + 0.25 KiB of LDS-DMA per v_mfma_f32_32x32x16_bf16.  With two co-resident
// workgroups per CU that is ~770 LDS cycles per 1024 matrix-pipe cycles before any bank conflict or barrier skew
// (MI355X_MICROARCH.md §LDS: ds_read_b128 = 4 cycles per wave-instruction, 256 B/clk), and PMC puts those kernels at
// 45-47 % MFMA-busy (profiles/r03_mfma_util.json).  The only way down is more MFMAs per fragment byte:
//   * a wave owns 64 pixels x 160 channels (TM = 2, TN = 5: 160 accumulator registers): (2 + 5) fragment reads per 10
//     MFMAs = 0.7 KiB per MFMA;
//   * the four waves of a workgroup stack along the PIXELS (256 pixels = 16 x 16) and all read the SAME 160-channel weight
//     tile: 20 KiB of LDS-DMA per tap for 160 MFMAs = 0.125 KiB per MFMA;
//   * => 0.83 KiB of LDS traffic per MFMA instead of 1.25: ~45 % LDS-port load at full matrix rate.
// The price: 145 KiB of LDS (three 20 KiB weight stages + two 41 KiB patches) and > 256 registers per lane — ONE workgroup
// per CU, one wave per SIMD (512 registers each).  Nothing co-resides, so the loop has to cover its own latencies: the
// weight ring runs two taps ahead (counted vmcnt), the next chunk's patch is requested at tap 0 and may stay in flight
// through tap 2.  N = 320 needs exactly two channel tiles (the 128-wide tiles of lconv3_kernel run three, the last half
// empty: 17 % of the MFMAs of the biggest launches), and the patch of a pixel tile is staged twice instead of three
// times (L2 -> LDS traffic, roofline.traffic).  Levels with fewer than one tile per CU cut the channel chunks into slices
// (ticket split-K of lean.h).
// Measured and rejected (profiles/r04_wide_cost_attribution.txt): moving every LDS-DMA request to a FIFTH "loader" wave (the
// four MFMA waves then execute only barrier + 28 fragment reads + 40 MFMAs per tap; all waves capped at 256 registers) —
// 62.7 vs 58.7 us on 8x64x64 320->320.  The attribution says why: MFMAs alone 0.64 us per tap, MFMAs + fragment reads 0.73,
// MFMAs + DMA (no reads) 0.65, all three 1.06 — whichever wave issues them, the LDS-DMA writes and the ds_read_b128 stream
// contend for the LDS port (112 KiB of reads + 25 KiB of DMA writes per tap and CU), and the MFMAs starve behind the reads.
// Replaces nn.Conv2d(k=3, pad=1) of ResBlock (reference openaimodel.py:183-187,218-231) where the output channels are a
// multiple of 160 and the map a multiple of 16 x 16.
#pragma once
#include <type_traits>

namespace wd {

using g8::buf_lds16;
using g8::OOB;
using g8::wait_vm;
using lg::C3Params;
using lg::raw_barrier;
using lg::wait_vm_upto;
using lg::wave_colstats;

struct WGeo {
  static constexpr int TW = 16, TH = 16, NW = 4;
  static constexpr int TM = 2, TN = 5, BN = TN * 32;         // per wave: 64 pixels (4 image rows of the tile) x 160 channels
  static constexpr int PW = TW + 2, PH = TH + 2, PROWS = PW * PH;
  static constexpr int PIECES = (PROWS + 7) / 8;             // 1 KiB LDS-DMA pieces (8 patch rows of 128 bytes)
  static constexpr int PATCH_BYTES = PIECES * 1024;
  static constexpr int W_BYTES = BN * ROW_BYTES;             // one tap's weight tile: 160 rows x 64 input channels
  static constexpr int NRING = 3;
  static constexpr int RING = NRING * W_BYTES + 2 * PATCH_BYTES;
  static constexpr int EROW = BN * 4;                        // one fp32 output row of a wave
  static constexpr int STAGING = NW * 32 * EROW;             // epilogue: one 32-pixel pass per wave at a time
  static constexpr int SMEM = RING > STAGING ? RING : STAGING;
  static_assert(SMEM <= 160 * 1024, "one workgroup per CU");
};

template <bool STATS>
__global__ void __launch_bounds__(256, 1) wconv3_kernel(const C3Params p) {
  using G = WGeo;
  constexpr int NW = G::NW, TM = G::TM, TN = G::TN, BN = G::BN, PW = G::PW, NR = G::NRING;
  constexpr int WP = (BN / 8) / NW;                          // weight pieces per wave and tap (5)
  constexpr int PP = (G::PIECES + NW - 1) / NW;              // patch pieces per wave and chunk (11, padded with duplicates)
  constexpr int EROW = G::EROW;
  static_assert((BN / 8) % NW == 0, "weight pieces split evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wring = smem;
  char* const patches = smem + NR * G::W_BYTES;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int hi = lane >> 5;
  const int l3 = lane >> 3;
  const int pslot = lane & 7;
  const int row0 = wave * (TM * 32);                         // first tile pixel of this wave (row-major in the 16 x 16 tile)
  const int swz_w = (l31 >> 1) & 7;
  const int w_frag_row = l31 * ROW_BYTES;

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
  const int y0 = ty * G::TH, x0 = (rt - ty * p.tiles_x) * G::TW;
  const int c0 = slice * p.ch_per;
  int c1 = c0 + p.ch_per;
  if (c1 > p.chunks) c1 = p.chunks;

  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.a), 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w), 0, p.w_bytes, 0x00020000);
  // patch rows: 16-byte slots XOR-swizzled on the row index with the halo pitch removed (lean.h lconv3_kernel: the rows one
  // ds_read_b128 lane group touches are 16 consecutive pixels of one image row — same tile width here)
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
    while (idx >= G::PIECES) idx -= NW;
    p_piece[i] = idx;
    const int prow = idx * 8 + l3;
    const int yy = prow / PW;
    const int xx = prow - yy * PW;
    const int gy = y0 + yy - 1, gx = x0 + xx - 1;
    const bool ok = (prow < G::PROWS) && ((unsigned)gy < (unsigned)p.H) && ((unsigned)gx < (unsigned)p.W);
    const int koff = (pslot ^ patch_swz(prow, yy)) * 8;
    p_voff[i] = ok ? (unsigned)(((((long long)b * p.H + gy) * p.W + gx) * p.C + koff) * 2) : OOB;
  }
  int a_prow[TM], a_py[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int ml = row0 + tm * 32 + l31;
    const int py = ml / G::TW, px = ml - py * G::TW;
    a_py[tm] = py;
    a_prow[tm] = py * PW + px;                               // patch row of this lane's pixel at tap (0, 0)
  }
  auto issue_w = [&](int st, int c, int tap) {
    const int soff = (tap * p.C + c * 64) * 2;
    char* wbuf = wring + st * G::W_BYTES;
#pragma unroll
    for (int i = 0; i < WP; ++i) buf_lds16(rsrc_w, wbuf + (wave + NW * i) * 1024, w_voff[i], soff);
  };
  auto issue_patch = [&](int buf, int c) {
    char* pbuf = patches + buf * G::PATCH_BYTES;
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

  // patch pieces of the NEXT chunk are requested two per tap (taps 0 .. 5), behind the tap's weight tile
  constexpr int PPT = 2, PTAPS = (PP + PPT - 1) / PPT;       // 11 pieces = 5 x 2 + 1
  static_assert(PTAPS <= 7, "the next patch is complete two taps before its chunk starts");
  auto issue_patch_part = [&](int buf, int c, int part) {
    char* pbuf = patches + buf * G::PATCH_BYTES;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int i = part * PPT + j;
      if (i < PP) buf_lds16(rsrc_a, pbuf + p_piece[i] * 1024, p_voff[i], c * 128);
    }
  };
  issue_patch(c0 & 1, c0);
#pragma unroll
  for (int j = 0; j < NR - 1; ++j) issue_w(j, c0, j);
  int st = 0;
  for (int c = c0; c < c1; ++c) {
    // every tap issues the same loads (the scheduler then sees nine straight-line tap bodies): behind the last chunk the
    // "next" weight tiles / patch are re-reads of this chunk into buffers nobody reads again
    const int cn = (c + 1 < c1) ? c + 1 : c;
    const char* pbuf = patches + (c & 1) * G::PATCH_BYTES;
    auto tap_body = [&](auto tap_c) {
      constexpr int tap = decltype(tap_c)::value;
      // loads younger than this tap's weight tile W(t), in issue order: the patch pieces of tap t - 2, W(t + 1), the patch
      // pieces of tap t - 1: they stay in flight, everything older has landed (the first chunk's patch is issued whole in
      // the prologue, ahead of W(0))
      constexpr auto pcount = [](int t) { const int tt = (t + 9) % 9; return tt < PTAPS ? ((tt + 1) * PPT <= PP ? PPT : PP - tt * PPT) : 0; };
      wait_vm<pcount(tap - 2) + WP + pcount(tap - 1)>();
      raw_barrier();
      const int dy = tap / 3, dx = tap - dy * 3;
      const char* wbuf = wring + st * G::W_BYTES;
      int arow[TM], aswz[TM];
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int prow = a_prow[t] + dy * PW + dx;
        arow[t] = prow * ROW_BYTES;
        aswz[t] = patch_swz(prow, a_py[t] + dy);
      }
      bf16x8_t fx[2][TM], fw[2][TN];
      auto read_ks = [&](int ks, int slot2) {
        if (UDT_DBG(p.dbg, 3)) return;                       // (measurement builds: no fragment reads)
#pragma unroll
        for (int t = 0; t < TM; ++t) fx[slot2][t] = lds_read_frag(pbuf + arow[t] + (((ks * 2 + hi) ^ aswz[t]) << 4));
        const int slot = ((ks * 2 + hi) ^ swz_w) << 4;
#pragma unroll
        for (int t = 0; t < TN; ++t) fw[slot2][t] = lds_read_frag(wbuf + w_frag_row + t * 32 * ROW_BYTES + slot);
      };
      read_ks(0, 0);
      // k-step by k-step: the (2 + 5) fragments of step ks + 1 are requested between the 10 MFMAs of step ks (one wave per
      // SIMD: nobody else fills the read latency); the LDS-DMA requests go between the MFMAs of the LAST step, where no
      // fragment reads compete for the issue slots (an LDS-DMA instruction costs 60 .. 180 cycles of issue depending on what
      // the phase carries, MI355X_MICROARCH.md "LDS-DMA piece issue cost")
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks + 1 < 4) read_ks(ks + 1, (ks + 1) & 1);
        if (ks == 3) {
          const int ahead = tap + NR - 1;
          int s2 = st + NR - 1;
          if (s2 >= NR) s2 -= NR;
          if (!UDT_DBG(p.dbg, 0)) {
            if (ahead < 9) issue_w(s2, c, ahead);
            else issue_w(s2, cn, ahead - 9);
          }
          if constexpr (tap < PTAPS) {
            if (!UDT_DBG(p.dbg, 1)) issue_patch_part((c + 1) & 1, cn, tap);
          }
        }
        if (UDT_DBG(p.dbg, 2)) {                             // (measurement builds: fragments kept alive, no MFMAs)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) asm volatile("" ::"v"(fx[ks & 1][tm]));
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) asm volatile("" ::"v"(fw[ks & 1][tn]));
        } else {
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32(fw[ks & 1][tn], fx[ks & 1][tm], acc[tm][tn]);
        }
      }
      // the order the scheduler is asked for (one straight-line region per tap)
      __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
#pragma unroll
        for (int i = 0; i < TM + TN; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
      }
      constexpr int NV = WP + pcount(tap);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - NV, 0);
      st = (st + 1 == NR) ? 0 : st + 1;
    };
    tap_body(std::integral_constant<int, 0>{}); tap_body(std::integral_constant<int, 1>{}); tap_body(std::integral_constant<int, 2>{});
    tap_body(std::integral_constant<int, 3>{}); tap_body(std::integral_constant<int, 4>{}); tap_body(std::integral_constant<int, 5>{});
    tap_body(std::integral_constant<int, 6>{}); tap_body(std::integral_constant<int, 7>{}); tap_body(std::integral_constant<int, 8>{});
  }
  // epilogue addressing (a row = one pixel x 160 channels; 20 lanes cover a row, 8 channels each, three rows per instruction)
  // and the residual rows of BOTH 32-pixel passes, requested before the ring drains: their latency overlaps the drain, the
  // split-K hand-off and the first pass (one wave per SIMD: 512 registers, 88 of them hold these rows)
  constexpr int CPR = TN * 4;                                // 8-channel groups per row (20)
  constexpr int RPI = 64 / CPR;                              // rows per instruction (3; lanes 60..63 idle)
  constexpr int NIT = (32 + RPI - 1) / RPI;                  // 11
  const int rl = lane / CPR;
  const int c8 = lane - rl * CPR;
  const int n = n0 + c8 * 8;
  const bool col_ok = (rl < RPI) && (n < p.N);
  long long mrow[TM][NIT];
  bool rok[TM][NIT];
  u32x4 rv[TM][NIT];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int row = i * RPI + rl;
      rok[tm][i] = col_ok && row < 32;
      const int ml = row0 + tm * 32 + (row < 32 ? row : 0);
      const int py = ml / G::TW, px = ml - py * G::TW;
      mrow[tm][i] = ((long long)b * p.H + (y0 + py)) * p.W + (x0 + px);
      u32x4 z = {0u, 0u, 0u, 0u};
      rv[tm][i] = z;
      if (p.res && rok[tm][i]) rv[tm][i] = *reinterpret_cast<const u32x4*>(p.res + mrow[tm][i] * p.ldr + n);
    }
  wait_vm<0>();                                              // (the trailing re-reads land before the ring becomes staging space)
  raw_barrier();

  if (p.splitk > 1) {
    f32x4* slab = reinterpret_cast<f32x4*>(p.slabs + (long long)unit * (G::TW * G::TH * BN));
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
      const f32x4* sl = reinterpret_cast<const f32x4*>(p.slabs + ((long long)tile * p.splitk + s) * (G::TW * G::TH * BN));
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

  // ---- epilogue: fp32 rows through this wave's own staging block (lean.h), one 32-pixel pass at a time
  char* const wl = smem + wave * (32 * EROW);
  f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
  if (col_ok) {
    if (p.bias) {
      b0 = *reinterpret_cast<const f32x4*>(p.bias + n);
      b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
    }
    if (p.rowvec) {                                          // one image per tile: the time-embedding row is a per-lane constant
      const float* rvp = p.rowvec + (long long)b * p.ldrv + n;
      const f32x4 r0 = *reinterpret_cast<const f32x4*>(rvp), r1 = *reinterpret_cast<const f32x4*>(rvp + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) b0[j] += r0[j], b1[j] += r1[j];
    }
  }
  float cs[8], cq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = cq[j] = 0.f;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    {
      const int row = l31;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int chunk = (tn * 8 + q * 2 + hi) ^ (row & 7);
          f32x4 v = {acc[tm][tn][q * 4 + 0], acc[tm][tn][q * 4 + 1], acc[tm][tn][q * 4 + 2], acc[tm][tn][q * 4 + 3]};
          *reinterpret_cast<f32x4*>(wl + row * EROW + chunk * 16) = v;
        }
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int row = i * RPI + rl;
      const int rr = row < 32 ? row : 0;
      const char* rp = wl + rr * EROW;
      const int sw = rr & 7;
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
          o[2 * j] += bf16_lo(rv[tm][i][j]);
          o[2 * j + 1] += bf16_hi(rv[tm][i][j]);
        }
      }
      if (rok[tm][i]) {
        u32x4 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
        *reinterpret_cast<u32x4*>(p.out + mrow[tm][i] * p.ldo + n) = pk;
        if constexpr (STATS) {                               // statistics of the values as stored (bf16-rounded)
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
    // one slot per wave pixel block (64 pixels of one image)
    const int slot = tile_m * NW + wave;
    wave_colstats<CPR, RPI>(wl, lane, cs, cq, p.colstats + ((long long)slot * p.N + n) * 2, col_ok);
  }
}

}  // namespace wd
