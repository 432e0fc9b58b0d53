// gemm9.h — 256x256-tile, phase-interleaved ("ping-pong") 8-wave GEMM / implicit-conv kernel (included by gemm.hip).
//
// Geometry: 8 waves as 2(M) x 4(N), each owning 128 x 64 of the 256 x 256 tile (4 x 2 MFMA 32x32 accumulators, 128
// VGPRs).  A K-tile (64 deep) lives in FOUR 16 KiB LDS regions — A0 A1 B0 B1 — where A<h> holds, for BOTH wave rows,
// the h-th 64-row half of each wave's 128 rows, and B<j> holds, for all four wave columns, the j-th 32-column half of
// each wave's 64 columns.  Two such buffers (128 KiB).
//
// A K-tile is computed in four PHASES, one 64x32 accumulator quadrant x K=64 (8 MFMAs) each:
//     P1: read A0,B0   MFMA(A0,B0)   stage B1(t+1)        P3: read A1   MFMA(A1,B1)   stage A0(t+2)
//     P2: read B1      MFMA(A0,B1)   stage A1(t+1)        P4:           MFMA(A1,B0)   stage B0(t+2)
// Every phase is  {ds_reads, LDS-DMA issue, counted vmcnt} -> s_barrier -> {MFMAs at raised priority} -> s_barrier.
// The four waves of wave-row 1 run ONE BARRIER behind the four of wave-row 0, and each SIMD hosts one wave of each
// row: while one wave of a SIMD issues MFMAs, the other issues its LDS reads and DMA requests, so the matrix pipe
// never waits for the load phase (cdna_hip_programming.md, "256^2 8-phase template").  A region is restaged no
// earlier than two phases after its last read, and read no earlier than one phase after the counted wait that
// retires its DMA; four half-tile requests (8 loads per wave) stay in flight across every barrier.
//
// The persistent / stream-K outer structure (ranges of (tile, K-tile) iterations, slab + flag publication, in-kernel
// finishing) and the fused epilogue are those of gemm8.h.
#pragma once

namespace g9 {

constexpr int NTHREADS = 512;
constexpr int BM = 256, BN = 256;
constexpr int HALF_BYTES = 128 * ROW_BYTES;      // 16 KiB: 128 LDS rows of one K-tile
constexpr int BUF_BYTES = 4 * HALF_BYTES;        // A0 A1 B0 B1
constexpr int SMEM_BYTES = 2 * BUF_BYTES;        // 128 KiB
constexpr int TM = 4, TN = 2;
constexpr int NTHREADS_EPI = NTHREADS;

using g8::buf_lds16;
using g8::OOB;
using g8::raw_barrier;
using g8::wait_vm;

// ---- fused epilogue, one 4-value accumulator group at a time -----------------------------------------------------
// plain / conv: row m, columns n..n+3.  The same function serves the register epilogue (unrolled, compile-time
// accumulator indices) and the slab epilogue (runtime loops over the parked partial tiles).
UDT_DEVINL void epi_store(const GemmParams& p, int batch, int m, int n, f32x4 acc4) {
  if (m >= p.M || n >= p.N) return;
  const int flags = p.flags;
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = acc4[r] * p.alpha;
  if (p.bias) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += bv[r];
  }
  if (p.rowvec) {
    const int b = m / p.rows_per_batch;
    const f32x4 rv = *reinterpret_cast<const f32x4*>(p.rowvec + (long long)b * p.ldrv + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += rv[r];
  }
  if (p.res) {
    const uint16_t* R = p.res + (long long)batch * p.sR;
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

// transposed store: rows m..m+3 (tokens of one batch element), output channel n -> out[b][n][tok..tok+3]
UDT_DEVINL void epi_store_trans(const GemmParams& p, int m, int n, f32x4 acc4) {
  if (m >= p.M || n >= p.N) return;
  const float bias = p.bias ? p.bias[n] : 0.f;
  const int b = m / p.rows_per_batch;
  const int tok = m - b * p.rows_per_batch;
  u32x2 pk = {pack_bf16x2(acc4[0] * p.alpha + bias, acc4[1] * p.alpha + bias),
              pack_bf16x2(acc4[2] * p.alpha + bias, acc4[3] * p.alpha + bias)};
  uint16_t* outT = reinterpret_cast<uint16_t*>(p.out);
  *reinterpret_cast<u32x2*>(outT + ((long long)b * p.N + n) * p.rows_per_batch + tok) = pk;
}

// GEGLU: value columns nx..nx+3 and their gates (nx+32..) -> out[m][no..no+3] = x * gelu(gate)
UDT_DEVINL void epi_store_geglu(const GemmParams& p, int batch, int m, int nx, int no, f32x4 xv, f32x4 gv) {
  if (m >= p.M || nx >= p.N) return;
  f32x4 bx = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
    bx = *reinterpret_cast<const f32x4*>(p.bias + nx);
    bg = *reinterpret_cast<const f32x4*>(p.bias + nx + 32);
  }
  float o[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = (xv[r] * p.alpha + bx[r]) * gelu_erf_f(gv[r] * p.alpha + bg[r]);
  uint16_t* out = reinterpret_cast<uint16_t*>(p.out) + (long long)batch * p.sO;
  u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
  *reinterpret_cast<u32x2*>(out + (long long)m * p.ldo + no) = pk;
}

// register epilogue of a wave's TM x TN accumulators at tile origin (m0 + row0, n0 + col0)
template <int TM_, int TN_, bool TRANS>
UDT_DEVINL void epilogue_regs(const GemmParams& p, f32x16 (&acc)[TM_][TN_], int batch, int mw, int nw, int lane) {
  const int l31 = lane & 31, hi = lane >> 5;
  const bool geglu = (p.flags & UDT_GEMM_GEGLU) != 0;
#pragma unroll
  for (int tm = 0; tm < TM_; ++tm)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!TRANS && geglu) {
        if constexpr (TN_ == 2) {
          f32x4 xv = {acc[tm][0][q * 4 + 0], acc[tm][0][q * 4 + 1], acc[tm][0][q * 4 + 2], acc[tm][0][q * 4 + 3]};
          f32x4 gv = {acc[tm][1][q * 4 + 0], acc[tm][1][q * 4 + 1], acc[tm][1][q * 4 + 2], acc[tm][1][q * 4 + 3]};
          epi_store_geglu(p, batch, mw + tm * 32 + l31, nw + q * 8 + hi * 4, (nw >> 1) + q * 8 + hi * 4, xv, gv);
        }
      } else {
#pragma unroll
        for (int tn = 0; tn < TN_; ++tn) {
          f32x4 v = {acc[tm][tn][q * 4 + 0], acc[tm][tn][q * 4 + 1], acc[tm][tn][q * 4 + 2], acc[tm][tn][q * 4 + 3]};
          if constexpr (TRANS) epi_store_trans(p, mw + tm * 32 + q * 8 + hi * 4, nw + tn * 32 + l31, v);
          else epi_store(p, batch, mw + tm * 32 + l31, nw + tn * 32 + q * 8 + hi * 4, v);
        }
      }
    }
}

// slab epilogue: the tile's partial accumulators were parked by this workgroup (own_slab) and by g_first..g_last in
// fragment order [(tm*TN+tn)*4+q][tid] float4; sum them group by group and run the same fused store
template <int TM_, int TN_, bool TRANS>
UDT_DEVINL void epilogue_slabs(const GemmParams& p, const float* own_slab, const float* slab_base, long long slab_floats,
                               int g_first, int g_last, int batch, int mw, int nw, int tid) {
  const int lane = tid & 63;
  const int l31 = lane & 31, hi = lane >> 5;
  const bool geglu = (p.flags & UDT_GEMM_GEGLU) != 0;
  // own_slab: this workgroup's partial; partners g_first..g_last parked theirs in the publish region
  const f32x4* own = reinterpret_cast<const f32x4*>(own_slab) + tid;
  const f32x4* base = reinterpret_cast<const f32x4*>(slab_base + (long long)g_first * slab_floats) + tid;
  const long long gstride = slab_floats / 4;
  const int np = g_last - g_first + 1;
  auto gather = [&](int grp) {
    const f32x4* s = base + (long long)grp * NTHREADS_EPI;
    f32x4 a = own[(long long)grp * NTHREADS_EPI];
    int i = 0;
    for (; i + 1 < np; i += 2) {
      const f32x4 b0 = s[(long long)i * gstride];
      const f32x4 b1 = s[(long long)(i + 1) * gstride];
#pragma unroll
      for (int r = 0; r < 4; ++r) a[r] += b0[r] + b1[r];
    }
    if (i < np) {
      const f32x4 b0 = s[(long long)i * gstride];
#pragma unroll
      for (int r = 0; r < 4; ++r) a[r] += b0[r];
    }
    return a;
  };
#pragma unroll 1
  for (int tm = 0; tm < TM_; ++tm)
#pragma unroll 2
    for (int q = 0; q < 4; ++q) {
      if (!TRANS && geglu) {
        if constexpr (TN_ == 2) {
          const f32x4 xv = gather((tm * TN_ + 0) * 4 + q);
          const f32x4 gv = gather((tm * TN_ + 1) * 4 + q);
          epi_store_geglu(p, batch, mw + tm * 32 + l31, nw + q * 8 + hi * 4, (nw >> 1) + q * 8 + hi * 4, xv, gv);
        }
      } else {
#pragma unroll
        for (int tn = 0; tn < TN_; ++tn) {
          const f32x4 v = gather((tm * TN_ + tn) * 4 + q);
          if constexpr (TRANS) epi_store_trans(p, mw + tm * 32 + q * 8 + hi * 4, nw + tn * 32 + l31, v);
          else epi_store(p, batch, mw + tm * 32 + l31, nw + tn * 32 + q * 8 + hi * 4, v);
        }
      }
    }
}

// MFMA with the accumulator pinned to AGPRs: the K loop then needs ~110 VGPRs (fragments + addresses) next to 128
// accumulator AGPRs, instead of asking the allocator to pack 8 x 16-register tuples among them (it spills)
UDT_DEVINL void mfma32a(f32x16& acc, bf16x8_t a, bf16x8_t b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

template <bool CONV, bool TRANS>
__global__ void __launch_bounds__(NTHREADS) gemm9_kernel(const g8::Params pp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const GemmParams& p = pp.g;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int hi = lane >> 5;
  const int l3 = lane >> 3;
  const int pslot = lane & 7;
  const int wr = wave >> 2;                      // wave row: 0 leads, 1 runs one barrier behind
  const int wc = wave & 3;
  const int row0 = wr * 128;
  const int col0 = wc * 64;
  const int swz = (l31 >> 1) & 7;

  const int g = range_index(blockIdx.x, p.G);
  long long it = (long long)g * p.iters_per_wg;
  long long it_end = it + p.iters_per_wg;
  if (it_end > p.total_iters) it_end = p.total_iters;
  if (it >= it_end) return;

  const int Ctot = p.C1 + p.C2;
  const int Hv = p.Hin << p.ups;
  const int Wv = p.Win << p.ups;

  // ---- fragment read addresses (bytes inside a buffer): region base + row*128 + swizzled 16-byte slot --------------
  // slot(ks) = ((ks*2 + hi) ^ swz) * 16
  int a_rd[4], b_rd[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int slot = ((ks * 2 + hi) ^ swz) << 4;
    a_rd[ks] = (wr * 64 + l31) * ROW_BYTES + slot;                   // + h*HALF_BYTES + t*32*ROW_BYTES
    b_rd[ks] = 2 * HALF_BYTES + (wc * 32 + l31) * ROW_BYTES + slot;  // + j*HALF_BYTES
  }

  // ---- staging state: each wave owns pieces 2*wave, 2*wave+1 (8 rows x 128 B each) of every half-tile ------------
  __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.a), 0, pp.a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w), 0, pp.w_bytes, 0x00020000);
  unsigned a_voff[2][2], w_voff[2][2];
  int a_iy0[2][2], a_ix0[2][2], a_pixb[2][2];
  int koff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 8 + l3;
    koff[i] = (pslot ^ ((r >> 1) & 7)) * 8;
  }

  auto prepare = [&](int batch, int m0, int n0) {
    if constexpr (!CONV) {
      rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.a + (long long)batch * p.sA), 0, pp.a_bytes,
                                                 0x00020000);
      rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w + (long long)batch * p.sW), 0, pp.w_bytes,
                                                 0x00020000);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 8 + l3;                       // LDS row inside the half-tile
        const int m = m0 + (r >> 6) * 128 + h * 64 + (r & 63);
        if constexpr (CONV) {
          const int hw = p.Hout * p.Wout;
          const int b = m / hw;
          const int rem = m - b * hw;
          const int oy = rem / p.Wout;
          const int ox = rem - oy * p.Wout;
          a_pixb[h][i] = b * p.Hin * p.Win;
          a_iy0[h][i] = (m < p.M) ? (oy * p.stride - p.pad_t) : -100000;
          a_ix0[h][i] = ox * p.stride - p.pad_l;
          a_voff[h][i] = 0;
        } else {
          a_voff[h][i] = (m < p.M) ? (unsigned)(((long long)m * p.lda + koff[i]) * 2) : OOB;
          a_iy0[h][i] = a_ix0[h][i] = a_pixb[h][i] = 0;
        }
      }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 8 + l3;
        const int n = n0 + (r >> 5) * 64 + j * 32 + (r & 31);
        w_voff[j][i] = (n < p.N) ? (unsigned)(((long long)n * p.ldw + koff[i]) * 2) : OOB;
      }
  };

  auto stage_a = [&](int b, int h, int kt) {
    char* dst = smem + b * BUF_BYTES + h * HALF_BYTES + wave * 2048;
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
      for (int i = 0; i < 2; ++i) {
        const int iy = a_iy0[h][i] + ky;
        const int ix = a_ix0[h][i] + kx;
        const bool ok = ((unsigned)iy < (unsigned)Hv) && ((unsigned)ix < (unsigned)Wv);
        const long long pix = (long long)a_pixb[h][i] + (long long)(iy >> p.ups) * p.Win + (ix >> p.ups);
        const uint16_t* gp = ok ? (src + pix * cs + cc + koff[i]) : p.zero;
        glds16(gp, dst + i * 1024);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) buf_lds16(rsrc_a, dst + i * 1024, a_voff[h][i], k0 * 2);
    }
  };
  auto stage_b = [&](int b, int j, int kt) {
    char* dst = smem + b * BUF_BYTES + (2 + j) * HALF_BYTES + wave * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i) buf_lds16(rsrc_w, dst + i * 1024, w_voff[j][i], kt * BK * 2);
  };

  // request the first six half-tiles of a segment [kt0, kt1): A0 B0 B1 A1 of kt0, then A0 B0 of kt0+1
  auto prologue = [&](int kt0, int kt1) {
    stage_a(0, 0, kt0);
    stage_b(0, 0, kt0);
    stage_b(0, 1, kt0);
    stage_a(0, 1, kt0);
    if (kt0 + 1 < kt1) {
      stage_a(1, 0, kt0 + 1);
      stage_b(1, 0, kt0 + 1);
    }
  };

  int tile = (int)(it / p.n_ktiles);
  int kt0 = (int)(it - (long long)tile * p.n_ktiles);
  int batch, m0, n0;
  decode_tile<BM, BN>(p, tile, batch, m0, n0);
  prepare(batch, m0, n0);
  {
    int kt1 = p.n_ktiles;
    if ((long long)(kt1 - kt0) > it_end - it) kt1 = kt0 + (int)(it_end - it);
    prologue(kt0, kt1);
  }

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

    // A0, B0 of kt0 landed (everything older than the newest 2 or 4 half-tile requests)
    if (kt0 + 1 < kt1) wait_vm<8>();
    else wait_vm<4>();
    raw_barrier();
    if (wr == 1) raw_barrier();                    // wave row 1 falls one barrier behind

    bf16x8_t af[2][4], bf0[4], bf1[4];

    // one K-tile = four phases; B = buffer holding K-tile kt
    auto ktile = [&](int B, int kt) {
      const char* buf = smem + B * BUF_BYTES;
      const bool has1 = kt + 1 < kt1;
      const bool has2 = kt + 2 < kt1;

      // ---------------- P1: quadrant (A0, B0) ----------------
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bf0[ks] = lds_read_frag(buf + b_rd[ks]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        af[0][ks] = lds_read_frag(buf + a_rd[ks]);
        af[1][ks] = lds_read_frag(buf + a_rd[ks] + 32 * ROW_BYTES);
      }
      if (has1) { stage_b(B ^ 1, 1, kt + 1); wait_vm<8>(); }         // B1(kt) landed
      else wait_vm<2>();
      __builtin_amdgcn_sched_barrier(0);
      raw_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if constexpr (TRANS) mfma32a(acc[t][0], af[t][ks], bf0[ks]);
          else mfma32a(acc[t][0], bf0[ks], af[t][ks]);
        }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      raw_barrier();

      // ---------------- P2: quadrant (A0, B1) ----------------
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bf1[ks] = lds_read_frag(buf + b_rd[ks] + HALF_BYTES);
      if (has1) { stage_a(B ^ 1, 1, kt + 1); wait_vm<8>(); }         // A1(kt) landed
      else wait_vm<0>();
      __builtin_amdgcn_sched_barrier(0);
      raw_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if constexpr (TRANS) mfma32a(acc[t][1], af[t][ks], bf1[ks]);
          else mfma32a(acc[t][1], bf1[ks], af[t][ks]);
        }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      raw_barrier();

      // ---------------- P3: quadrant (A1, B1) ----------------
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        af[0][ks] = lds_read_frag(buf + a_rd[ks] + HALF_BYTES);
        af[1][ks] = lds_read_frag(buf + a_rd[ks] + HALF_BYTES + 32 * ROW_BYTES);
      }
      if (has2) stage_a(B, 0, kt + 2);
      __builtin_amdgcn_sched_barrier(0);
      raw_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if constexpr (TRANS) mfma32a(acc[2 + t][1], af[t][ks], bf1[ks]);
          else mfma32a(acc[2 + t][1], bf1[ks], af[t][ks]);
        }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      raw_barrier();

      // ---------------- P4: quadrant (A1, B0) ----------------
      if (has2) { stage_b(B, 0, kt + 2); wait_vm<8>(); }             // A0, B0 of kt+1 landed
      else if (has1) wait_vm<4>();
      __builtin_amdgcn_sched_barrier(0);
      raw_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if constexpr (TRANS) mfma32a(acc[2 + t][0], af[t][ks], bf0[ks]);
          else mfma32a(acc[2 + t][0], bf0[ks], af[t][ks]);
        }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      raw_barrier();
    };

    int bsel = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
      ktile(bsel, kt);
      bsel ^= 1;
    }
    if (wr == 0) raw_barrier();                    // wave row 0 waits for row 1: every wave is done with the LDS

    // opaque copies: keeps every lane-derived epilogue address out of the K loop's live range (the accumulators,
    // fragments and staging offsets fill the register file there)
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63;

    const bool full = (kt0 == 0) && (kt1 == p.n_ktiles);
    const bool publish = (kt0 > 0);                       // head segment of a tile another workgroup started
    const int cur_tile = tile, cur_batch = batch, cur_m0 = m0, cur_n0 = n0;
    it += kt1 - kt0;
    const bool more = it < it_end;
    if (more) {
      tile = (int)(it / p.n_ktiles);
      kt0 = (int)(it - (long long)tile * p.n_ktiles);
      decode_tile<BM, BN>(p, tile, batch, m0, n0);
      prepare(batch, m0, n0);
      int nk1 = p.n_ktiles;
      if ((long long)(nk1 - kt0) > it_end - it) nk1 = kt0 + (int)(it_end - it);
      prologue(kt0, nk1);
    }

    if (!full) {
      // park this segment's accumulators (publisher AND finisher: the finisher then reduces from the slabs, so the
      // accumulators are never modified outside the K loop)
      // slab g is read by the finisher of an EARLIER tile at an unknown time, so a finisher parks into slab G + g
      float* own_slab = pp.slab_base + (long long)((publish ? 0 : p.G) + g) * (BM * BN);
      f32x4* slab = reinterpret_cast<f32x4*>(own_slab);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v = {acc[tm][tn][q * 4 + 0], acc[tm][tn][q * 4 + 1], acc[tm][tn][q * 4 + 2],
                       acc[tm][tn][q * 4 + 3]};
            store16_sc1(slab + ((tm * TN + tn) * 4 + q) * NTHREADS + tid_e, v);
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // every storing wave drains its stores
      __syncthreads();
      if (publish) {
        if (tid_e == 0) {
          // slab stores were write-through (sc1) and are drained: no L2 write-back fence needed
          __hip_atomic_store(pp.flags + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
        // this workgroup owns the start of the tile: wait for the partners, reduce all slabs, fused store
        const long long tile_end = ((long long)cur_tile + 1) * p.n_ktiles;
        const int g_last = (int)((tile_end - 1) / p.iters_per_wg);
        if (tid_e == 0) {
          for (int pg = g + 1; pg <= g_last; ++pg) {
            int spins = 0;
            while (__hip_atomic_load(pp.flags + pg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
              __builtin_amdgcn_s_sleep(8);
              if (++spins > g8::SPIN_LIMIT) {
                __hip_atomic_store(pp.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
              }
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        epilogue_slabs<TM, TN, TRANS>(p, own_slab, pp.slab_base, (long long)BM * BN, g + 1, g_last, cur_batch,
                                      cur_m0 + row0, cur_n0 + col0, tid_e);
        __syncthreads();
        if (tid_e == 0)
          for (int pg = g + 1; pg <= g_last; ++pg)
            __hip_atomic_store(pp.flags + pg, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      epilogue_regs<TM, TN, TRANS>(p, acc, cur_batch, cur_m0 + row0, cur_n0 + col0, lane_e);
    }
    if (!more) break;
    // the epilogue's stores are younger than the next segment's requests: drain everything, then count from zero
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

}  // namespace g9
