// attention.hip — attention kernels for gfx950.
//
// udt_attn_fwd: flash attention forward, head_dim 64 (UNet self-attention; replaces
//   xformers.ops.memory_efficient_attention at sgm/modules/attention.py:246).
//   One workgroup = 128 queries of one (batch, head); 4 waves x 32 queries.  K and V^T tiles of 64 keys
//   arrive by 16-byte LDS-DMA into a double-buffered, XOR-swizzled LDS image (same scheme as gemm.hip).
//   Scores are computed "swapped": S^T = K·Q^T with the K fragment as MFMA A operand, so every lane owns
//   ONE query column and 32 of the tile's 64 keys -> the online-softmax row max / row sum are lane-local
//   plus one cross-half shuffle, and the rescale factor is a per-lane scalar for the O^T accumulator
//   (O^T = V^T·P^T, V^T fragment as A operand).  The P fragment is fed to the second MFMA in the key
//   order the first MFMA produced it (keys 4*hi+{0..3} and 8+4*hi+{0..3} of each 16-key step); the V^T
//   fragment is read from LDS in the matching order (two 8-byte reads), so no cross-lane permute is needed.
//
// udt_xattn_fwd: short-context attention (L <= 16 keys; the text cross-attention t_attn over the 12
//   character embeddings, sgm/modules/attention.py:152-172, and the LabelEncoder's 12-token
//   self-attention).  VALU kernel: one lane per (query, head), K/V of the (batch, head) broadcast from LDS,
//   optional write-out of the softmax probabilities ("attn_map_cache", attention.py:165-169).
//
// udt_attn512_fwd: flash attention forward for ONE head of 512 dims (the AutoencoderKL mid-block attention, model.py:236-260):
//   the head dimension is split over the four waves of a workgroup (attn_d512_body).
//
// udt_softmax_rows: in-place row softmax (the VAE attention's earlier GEMM -> softmax -> GEMM form, kept for A/B).
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <atomic>

namespace {

struct AttnParams {
  const uint16_t* q;
  const uint16_t* k;
  const uint16_t* vt;
  uint16_t* o;
  const uint16_t* zero;
  int heads, nq, nk;
  int ldq, ldk, ldvt, ldo;
  long long sq, sk, svt, so;
  float scale_log2e;
  // attn_d64_v2_kernel only (udt_attn_rowv_q8_fwd, BASELINE config #5): O again as an MX8 activation (common.h) for the e4m3 to_out GEMM
  uint8_t* q8_out;         // [batch * nq, ld_q8] e4m3, columns h * 64 + d
  uint32_t* q8_scale;      // [heads * 64 / 128][batch * nq]
  int ld_q8;
  long long q8_rows;       // batch * nq
  // attn_d64_v2_kernel: 1-D grid of G = units rounded up to 8 workgroups; unit = (batch * heads + head) * qtiles + query tile
  int qtiles, units, G;
};

constexpr int KV_TILE = 64;
constexpr int TILE_BYTES = 64 * 128;   // 64 rows x 128 B (K tile, and V^T tile)

// VROW = true: V is row-major like K (row (b, tok) at vt + b*svt + tok*ldvt + h*64 — the q|k|v output of ONE GEMM, no
// transposed-epilogue GEMM in front of the kernel): the V tile is staged exactly like the K tile, and the V^T fragments
// of the P·V step come from ds_read_b64_tr_b16, gfx950's LDS transpose read — each 16-lane group passes the addresses of a
// [4 keys][16 dims] block (lane i: key i/4, dims 4*(i%4)..+3) and lane i receives dim i of the four keys, which is the
// A-operand order the swapped S^T left P in (layout verified by tools/probes/tr_read.cpp).
template <bool VROW>
__global__ void __launch_bounds__(256, 4) attn_d64_kernel(const AttnParams p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];   // [buf][K | VT]
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int hi = lane >> 5;
  const int bh = blockIdx.y;
  const int b = bh / p.heads;
  const int h = bh - b * p.heads;

  const uint16_t* __restrict__ Q = p.q + (long long)b * p.sq + h * 64;
  const uint16_t* __restrict__ K = p.k + (long long)b * p.sk + h * 64;
  const uint16_t* __restrict__ VT = p.vt + (long long)b * p.svt + (VROW ? (long long)h * 64 : (long long)h * 64 * p.ldvt);
  uint16_t* __restrict__ O = p.o + (long long)b * p.so + h * 64;

  // this lane's query
  const int qi = blockIdx.x * 128 + wave * 32 + l31;
  const bool qok = qi < p.nq;

  // Q fragments (B operand of S^T = K Q^T): 8 consecutive d at 16*ks + 8*hi
  bf16x8_t qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const uint16_t* g = qok ? (Q + (long long)qi * p.ldq + ks * 16 + hi * 8) : p.zero;
    qf[ks] = *reinterpret_cast<const bf16x8_t*>(g);
  }

  // staging: each wave moves 2 K pieces and 2 V^T pieces (1 KiB each) per tile
  const int l3 = lane >> 3;
  const int pslot = lane & 7;
  int st_row[2], st_koff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    st_row[i] = (wave * 2 + i) * 8 + l3;
    st_koff[i] = (pslot ^ ((st_row[i] >> 1) & 7)) * 8;
  }
  auto stage = [&](int buf, int kt) {
    char* kbuf = smem + buf * 2 * TILE_BYTES;
    char* vbuf = kbuf + TILE_BYTES;
    const int key0 = kt * KV_TILE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int key = key0 + st_row[i];
      const uint16_t* g = (key < p.nk) ? (K + (long long)key * p.ldk + st_koff[i]) : p.zero;
      glds16(g, kbuf + (wave * 2 + i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if constexpr (VROW) {
        const int key = key0 + st_row[i];
        const uint16_t* g = (key < p.nk) ? (VT + (long long)key * p.ldvt + st_koff[i]) : p.zero;
        glds16(g, vbuf + (wave * 2 + i) * 1024);
      } else {
        const int kcol = key0 + st_koff[i];          // first of 8 keys of this 16-byte chunk
        const uint16_t* g = (kcol < p.nk) ? (VT + (long long)st_row[i] * p.ldvt + kcol) : p.zero;
        glds16(g, vbuf + (wave * 2 + i) * 1024);
      }
    }
  };
  // VROW: per-lane pieces of the transpose-read addresses.  Block of 16-key step s4, dim tile dt: keys 16*s4 + 4*hi (+8)
  // .. + 3, dims dt*32 + 16*g1 .. + 15 (g1 = second 16-lane group of the half-wave); this lane points at key + (i >> 2),
  // dims + 4 * (i & 3).  The tile's XOR swizzle ((row >> 1) & 7 on the 16-byte chunk) does not depend on s4.
  const int tr_i = lane & 15, tr_g1 = (lane >> 4) & 1;
  const int tr_row = 4 * hi + (tr_i >> 2);
  const int tr_swz = (tr_row >> 1) & 7;                       // rows + 8 flip bit 2 of it
  const int tr_chunk = 2 * tr_g1 + ((tr_i & 3) >> 1);         // + 4 * dt
  const int tr_base = tr_row * 128 + (tr_i & 1) * 8;

  f32x16 o_acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
  float m_run = -INFINITY;   // running max of (score * scale * log2e)
  float l_run = 0.f;         // this lane's partial row sum

  const int swz = (l31 >> 1) & 7;
  const int frag_row = l31 * 128;
  const int ntiles = (p.nk + KV_TILE - 1) / KV_TILE;
  const float c = p.scale_log2e;

  stage(0, 0);
  for (int kt = 0; kt < ntiles; ++kt) {
    const int cur = kt & 1;
    wait_vmcnt0();
    __syncthreads();
    if (kt + 1 < ntiles) stage(cur ^ 1, kt + 1);
    const char* kbuf = smem + cur * 2 * TILE_BYTES;
    const char* vbuf = kbuf + TILE_BYTES;

    // ---- S^T[64 keys x 32 queries] -----------------------------------------------------------------
    f32x16 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int slot = ((ks * 2 + hi) ^ swz) << 4;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8_t kf = lds_read_frag(kbuf + frag_row + t * 32 * 128 + slot);
        s[t] = mfma32(kf, qf[ks], s[t]);
      }
    }
    // key of s[t][reg] = kt*64 + t*32 + 8*(reg>>2) + 4*hi + (reg&3)
    if (kt * KV_TILE + KV_TILE > p.nk) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * KV_TILE + t * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
          if (key >= p.nk) s[t][r] = -INFINITY;
        }
    }

    // ---- online softmax (per lane = per query) ------------------------------------------------------
    float mx = s[0][0];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
    mx = half_swap_max(mx);
    const float m_new = fmaxf(m_run, mx * c);
    const float alpha = fast_exp2(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = fast_exp2(s[t][r] * c - m_new);
        s[t][r] = pv;
        psum += pv;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o_acc[i][r] *= alpha;

    // ---- O^T += V^T · P^T ------------------------------------------------------------------------------
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {     // 16-key steps
      const int t = s4 >> 1;
      const int half = s4 & 1;
      u32x4 pk;
      pk[0] = pack_bf16x2(s[t][half * 8 + 0], s[t][half * 8 + 1]);
      pk[1] = pack_bf16x2(s[t][half * 8 + 2], s[t][half * 8 + 3]);
      pk[2] = pack_bf16x2(s[t][half * 8 + 4], s[t][half * 8 + 5]);
      pk[3] = pack_bf16x2(s[t][half * 8 + 6], s[t][half * 8 + 7]);
      const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pk);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        u32x2 lo, hh;
        if constexpr (VROW) {
          const unsigned blk = lds_offset(vbuf) + tr_base;
          const int c = dt * 4 + tr_chunk;
          lo = lds_tr16_b64(blk + ((c ^ tr_swz) << 4), s4 * 2048);
          hh = lds_tr16_b64(blk + ((c ^ tr_swz ^ 4) << 4), s4 * 2048 + 1024);
          lds_tr_wait(lo, hh);
        } else {
          const char* row = vbuf + (dt * 32 + l31) * 128;
          lo = *reinterpret_cast<const u32x2*>(row + (((2 * s4) ^ swz) << 4) + 8 * hi);
          hh = *reinterpret_cast<const u32x2*>(row + (((2 * s4 + 1) ^ swz) << 4) + 8 * hi);
        }
        u32x4 vv = {lo[0], lo[1], hh[0], hh[1]};
        o_acc[dt] = mfma32(__builtin_bit_cast(bf16x8_t, vv), pf, o_acc[dt]);
      }
    }
  }

  // ---- epilogue: O[q][d] = O^T / l ------------------------------------------------------------------------
  const float l_tot = half_swap_sum(l_run);
  const float inv = 1.0f / l_tot;
  if (qok) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int d = dt * 32 + qd * 8 + hi * 4;
        u32x2 pk = {pack_bf16x2(o_acc[dt][qd * 4 + 0] * inv, o_acc[dt][qd * 4 + 1] * inv),
                    pack_bf16x2(o_acc[dt][qd * 4 + 2] * inv, o_acc[dt][qd * 4 + 3] * inv)};
        *reinterpret_cast<u32x2*>(O + (long long)qi * p.ldo + d) = pk;
      }
  }
}

// ---------------------------------------------------------------------------------------------------
// attn_d64_v2_kernel (round 3): the row-major-V flash kernel with the loop overheads removed.  Same arithmetic and the same
// fragment orders as attn_d64_kernel<true> (128 queries per workgroup, 4 waves x 32 queries, swapped S^T = K Q^T, V^T
// fragments by LDS transpose reads), but
//   * a THREE-stage K|V ring filled two tiles ahead, counted `s_waitcnt vmcnt` + a raw barrier per 64-key tile (the
//     first kernel drained vmcnt(0) and had one tile in flight: every tile paid the L2 latency of its successor);
//   * every LDS address is a per-lane base computed once + an immediate (the transpose reads cost 3 VALU ops each before);
//   * the online-softmax rescale of O is skipped while the running maximum grows by less than 2^8 (wave-uniform test;
//     P then stays <= 256, exact in fp32 and harmless for the bf16 P fragments — the normaliser l sees the same P);
//   * MFMA groups run at raised wave priority (the other waves of the SIMD are in their softmax / read phases).
// 48 KiB of LDS -> three workgroups per CU.
// LDS: a ring of THREE K tiles and a ring of TWO V tiles (40 KiB -> four workgroups per CU; round 5 — it was three stages of K | V
// = 48 KiB and three workgroups): K is requested two tiles ahead, V one tile ahead (a V tile is not needed before the second half of
// its step, and a step is longer than the DMA round trip).
constexpr int A2_KRING = 3, A2_VRING = 2;
constexpr int A2_VBASE = A2_KRING * TILE_BYTES;
constexpr int A2_SMEM = (A2_KRING + A2_VRING) * TILE_BYTES;
// cost attribution (measurement builds only: -DUDT_MEASURE -DA2_HALF_MFMA; WRONG results): HALF of the QK^T and P V MFMAs and of the
// K / V fragment reads that feed them — what e4m3 K / V / P operands (v_mfma_scale_f32_32x32x64_f8f6f4: half the matrix-pipe time and
// half the LDS bytes per tile) could save at most, with the softmax arithmetic unchanged (profiles/r05_attn_fp8_bound.txt)
#if defined(UDT_MEASURE) && defined(A2_HALF_MFMA)
constexpr int A2_KS = 2, A2_DT = 1;
#else
constexpr int A2_KS = 4, A2_DT = 2;
#endif
constexpr float A2_DEFER = 8.0f;

__global__ void __launch_bounds__(256, 4) attn_d64_v2_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem2[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int hi = lane >> 5;
  // XCD-aware unit order: the hardware deals consecutive workgroups round-robin to the 8 XCDs, each with an L2 of its own.  With a
  // (query tile, batch * heads) grid every XCD works on every (batch, head) at once and the K / V of ~32 of them (0.5 MiB each at
  // 4096 keys) compete for its 4 MiB; range_index gives XCD x the contiguous unit range [x G / 8, (x + 1) G / 8): the query tiles of ONE
  // (batch, head) run together on ONE XCD and stream its K / V out of that L2.
  const int unit = range_index(blockIdx.x, p.G);
  if (unit >= p.units) return;
  const int bh = unit / p.qtiles;
  const int qt = unit - bh * p.qtiles;
  const int b = bh / p.heads;
  const int h = bh - b * p.heads;
  const uint16_t* __restrict__ Q = p.q + (long long)b * p.sq + h * 64;
  const uint16_t* __restrict__ K = p.k + (long long)b * p.sk + h * 64;
  const uint16_t* __restrict__ V = p.vt + (long long)b * p.svt + h * 64;
  uint16_t* __restrict__ O = p.o + (long long)b * p.so + h * 64;
  const int qi = qt * 128 + wave * 32 + l31;
  const bool qok = qi < p.nq;
  bf16x8_t qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const uint16_t* g = qok ? (Q + (long long)qi * p.ldq + ks * 16 + hi * 8) : p.zero;
    qf[ks] = *reinterpret_cast<const bf16x8_t*>(g);
  }
  // staging: 2 K pieces + 2 V pieces (1 KiB each) per wave and tile, through buffer descriptors (keys past nk read zeros)
  const int l3 = lane >> 3, pslot = lane & 7;
  const unsigned kbytes = (unsigned)(((long long)(p.nk - 1) * p.ldk + 64) * 2);
  const unsigned vbytes = (unsigned)(((long long)(p.nk - 1) * p.ldvt + 64) * 2);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(K), 0, kbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(V), 0, vbytes, 0x00020000);
  unsigned kvo[2], vvo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 8 + l3;
    const int koff = (pslot ^ ((row >> 1) & 7)) * 8;
    kvo[i] = (unsigned)(((long long)row * p.ldk + koff) * 2);
    vvo[i] = (unsigned)(((long long)row * p.ldvt + koff) * 2);
  }
  const int kstep = KV_TILE * p.ldk * 2, vstep = KV_TILE * p.ldvt * 2;        // bytes per 64-key tile
  auto stage_k = [&](int kt) __attribute__((always_inline)) {
    char* kbuf = smem2 + (kt % A2_KRING) * TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (__attribute__((address_space(3))) void*)(kbuf + (wave * 2 + i) * 1024), 16, kvo[i], kt * kstep, 0, 0);
  };
  auto stage_v = [&](int kt) __attribute__((always_inline)) {
    char* vbuf = smem2 + A2_VBASE + (kt % A2_VRING) * TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void*)(vbuf + (wave * 2 + i) * 1024), 16, vvo[i], kt * vstep, 0, 0);
  };
  // per-lane LDS offsets, fixed for the whole kernel
  const int swz = (l31 >> 1) & 7;
  int koff_l[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff_l[ks] = l31 * 128 + (((ks * 2 + hi) ^ swz) << 4);
  const int tr_i = lane & 15, tr_g1 = (lane >> 4) & 1;
  const int tr_row = 4 * hi + (tr_i >> 2);
  const int tr_swz = (tr_row >> 1) & 7;
  const int tr_chunk = 2 * tr_g1 + ((tr_i & 3) >> 1);
  const int tr_base = tr_row * 128 + (tr_i & 1) * 8;
  int voff_a[2], voff_b[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    const int c = dt * 4 + tr_chunk;
    voff_a[dt] = tr_base + ((c ^ tr_swz) << 4);                 // (relative to the V tile)
    voff_b[dt] = tr_base + 1024 + ((c ^ tr_swz ^ 4) << 4);
  }

  f32x16 o_acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int ntiles = (p.nk + KV_TILE - 1) / KV_TILE;
  const float c = p.scale_log2e;
  // request order: K(0), V(0), K(1) | step kt: V(kt + 1), K(kt + 2) — at the top of step kt everything but the two youngest loads
  // (K(kt + 1)) has to be there
  stage_k(0);
  stage_v(0);
  if (ntiles > 1) stage_k(1);
  int ks3 = 0;                                             // kt % 3
  for (int kt = 0; kt < ntiles; ++kt) {
    if (kt + 1 < ntiles) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");       // K(kt + 1) stays in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                               // every wave is past step kt - 1: its K and V tiles are free
    asm volatile("" ::: "memory");
    if (kt + 1 < ntiles) stage_v(kt + 1);
    if (kt + 2 < ntiles) stage_k(kt + 2);
    const char* buf = smem2 + ks3 * TILE_BYTES;            // this step's K tile
    unsigned tra[2], trb[2];                               // this tile's transposed-read addresses (lds_tr16_b64)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      tra[dt] = lds_offset(smem2) + A2_VBASE + (kt & 1) * TILE_BYTES + voff_a[dt];
      trb[dt] = lds_offset(smem2) + A2_VBASE + (kt & 1) * TILE_BYTES + voff_b[dt];
    }
    f32x16 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
    bf16x8_t kf[4][2];
#pragma unroll
    for (int ks = 0; ks < A2_KS; ++ks)
#pragma unroll
      for (int t = 0; t < 2; ++t) kf[ks][t] = lds_read_frag(buf + koff_l[ks] + t * 32 * 128);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < A2_KS; ++ks)
#pragma unroll
      for (int t = 0; t < 2; ++t) s[t] = mfma32(kf[ks][t], qf[ks], s[t]);
    __builtin_amdgcn_s_setprio(0);
    if (kt * KV_TILE + KV_TILE > p.nk) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * KV_TILE + t * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
          if (key >= p.nk) s[t][r] = -INFINITY;
        }
    }
    float mx = s[0][0];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
    mx = half_swap_max(mx);
    const float mxs = mx * c;
    if (!__all(mxs - m_run <= A2_DEFER)) {                  // wave-uniform: some query's maximum grew by more than 2^8
      const float m_new = fmaxf(m_run, mxs);
      const float alpha = fast_exp2(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[i][r] *= alpha;
    }
    // exponent arguments and the row sum on PACKED fp32 ops (v_pk_fma_f32 / v_pk_add_f32: two elements per instruction at
    // the scalar rate; the loop is bound by its VALU work — 32 quarter-rate v_exp_f32 plus ~100 other instructions per
    // 16 MFMAs — not by the matrix pipe)
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    const f32x2v c2 = {c, c}, m2 = {m_run, m_run};
    f32x2v ps2 = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2v e = {s[t][r], s[t][r + 1]};
        e = __builtin_elementwise_fma(e, c2, -m2);
        f32x2v pv = {fast_exp2(e[0]), fast_exp2(e[1])};
        s[t][r] = pv[0];
        s[t][r + 1] = pv[1];
        ps2 += pv;
      }
    l_run += ps2[0] + ps2[1];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int t = s4 >> 1, half = s4 & 1;
      u32x4 pk;
      pk[0] = pack_bf16x2(s[t][half * 8 + 0], s[t][half * 8 + 1]);
      pk[1] = pack_bf16x2(s[t][half * 8 + 2], s[t][half * 8 + 3]);
      pk[2] = pack_bf16x2(s[t][half * 8 + 4], s[t][half * 8 + 5]);
      pk[3] = pack_bf16x2(s[t][half * 8 + 6], s[t][half * 8 + 7]);
      const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pk);
      u32x4 vv[2];
      {
        u32x2 lo[2], hh[2];
#pragma unroll
        for (int dt = 0; dt < A2_DT; ++dt) {
          lo[dt] = lds_tr16_b64(tra[dt], s4 * 2048);
          hh[dt] = lds_tr16_b64(trb[dt], s4 * 2048);
        }
        if constexpr (A2_DT == 2) lds_tr_wait(lo[0], hh[0], lo[1], hh[1]); else lds_tr_wait(lo[0], hh[0]);
#pragma unroll
        for (int dt = 0; dt < A2_DT; ++dt) vv[dt] = u32x4{lo[dt][0], lo[dt][1], hh[dt][0], hh[dt][1]};
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int dt = 0; dt < A2_DT; ++dt) o_acc[dt] = mfma32(__builtin_bit_cast(bf16x8_t, vv[dt]), pf, o_acc[dt]);
      __builtin_amdgcn_s_setprio(0);
    }
    ks3 = ks3 + 1 == A2_KRING ? 0 : ks3 + 1;
  }
  const float l_tot = half_swap_sum(l_run);
  const float inv = 1.0f / l_tot;
  if (qok) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int d = dt * 32 + qd * 8 + hi * 4;
        u32x2 pk = {pack_bf16x2(o_acc[dt][qd * 4 + 0] * inv, o_acc[dt][qd * 4 + 1] * inv),
                    pack_bf16x2(o_acc[dt][qd * 4 + 2] * inv, o_acc[dt][qd * 4 + 3] * inv)};
        *reinterpret_cast<u32x2*>(O + (long long)qi * p.ldo + d) = pk;
      }
  }
  if (p.q8_out) {
    // a query's 32 dims of tile dt are one MX block: 16 of them in this lane, 16 in lane ^ 32 (every lane runs the exchange)
    const long long m = (long long)b * p.nq + qi;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = o_acc[dt][r] * inv;
      uint32_t q[4], sb;
      mx8_quant_acc16(v, q, sb);
      if (qok) {
        const int blk = h * 2 + dt;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) *reinterpret_cast<uint32_t*>(p.q8_out + m * p.ld_q8 + blk * 32 + qd * 8 + hi * 4) = q[qd];
        if (hi == 0) reinterpret_cast<uint8_t*>(p.q8_scale)[((long long)(blk >> 2) * p.q8_rows + m) * 4 + (blk & 3)] = (uint8_t)sb;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------
// attn_d64_mx8_kernel (round 5, BASELINE config #5's "fp8 attention"): the self-attention of attn_d64_v2_kernel on e4m3 operands.
// q, k, v are column ranges of ONE MX8 activation (the q|k|v projection's emitting epilogue, udt_gemm_desc.q8_out): q and k carry an
// E8M0 scale per (token, 32 head dims), v was written with one fixed multiplier (q8_fixed_col / q8_fixed_mul) because P V contracts
// over KEYS.  Per 64-key tile and wave (32 queries):
//   S^T = K Q^T   : 2 x v_mfma_scale_f32_32x32x64_f8f6f4 (one per 32 keys; the 64 head dims are the instruction's two scale blocks)
//   softmax       : maxima in fp32 with the deferred rescale of attn_d64_v2_kernel; the numerators straight to e4m3 bytes (below)
//   O^T += V^T P^T: 2 x the same instruction (one per 32 output dims; the 64 keys of the tile are its K) + 1 for the row sums
// i.e. 5 matrix instructions of 64 cycles instead of 16 of 32, half the LDS bytes per fragment and per staged tile.
// Two layout tricks make the operands fall into place (layouts measured by tools/probes/mx_*.cpp and tr_read_b8.cpp):
//   * the K rows of a 32-key fragment are PERMUTED (fragment lane i holds key 16 ((i >> 2) & 1) + 4 (i >> 3) + (i & 3)), so that the
//     16 scores a half-wave ends up with per 32-key tile are 16 CONSECUTIVE keys, in register order;
//   * the instruction wants, per lane (d, h), 16 key bytes of block 0 and 16 of block 1: with the permutation these are keys
//     16 h .. 16 h + 15 of the two 32-key halves of the tile — two ds_read_b64_tr_b8 each out of the row-major V tile (a 16-lane
//     group turns an 8-key x 16-dim byte block into 16 dims x 8 keys), and P's bytes are the lane's own scores in register order.
struct AttnMx8Params {
  const uint8_t* x8;       // [batch * n, ld8] e4m3: q at column h * 64, k at C + h * 64, v at 2 C + h * 64
  const uint32_t* sc;      // [ld8 / 128 (rounded up)][batch * n] block scales (bytes of the v third: unused)
  uint16_t* o;             // bf16 [batch * n, ldo], column h * 64 + d
  int heads, n, ld8, ldo, C;
  long long rows;          // batch * n
  float scale_log2e, v_inv;
  uint8_t* q8_out;         // optional: O again as an MX8 activation (as attn_d64_v2_kernel)
  uint32_t* q8_scale;
  int ld_q8;
  int qtiles, units, G;    // 1-D grid as attn_d64_v2_kernel
};
constexpr int A8_TILE = 64 * 64;                 // 64 keys x 64 bytes
constexpr int A8_STAGE = 2 * A8_TILE + 256;      // K tile + V tile + the K tile's scale dwords
constexpr int A8_NST = 3;
// The softmax numerators are produced as e4m3 BYTES without an exponential: an e4m3 bit pattern read as an integer is a
// piecewise-linear log2 scale (byte = 8 (log2 p + 7) + the mantissa's 8 (2^f - 1) ~ 8 f), so
//     byte(p) = rne(8 (s c - m_run) + 8 (A8_OFF + 7) + A8_ADJ)          — one (packed) FMA + one v_cvt_pk_u8_f32 per score
// (the conversion rounds to nearest even and saturates at 0 / 255; -inf and NaN give 0) instead of FMA + v_exp_f32 + add (row sum) +
// v_cvt_pk_fp8_f32 — measured issue cost per wave64 instruction relative to v_fma_f32: v_exp_f32 1.7, v_cvt_pk_fp8_f32 1.8 (two
// scores), v_cvt_pk_u8_f32 1.1, v_max3_f32 1.1 (profiles/r05_probe_cvt_pk_u8.txt).
// The row sum comes out of the matrix pipe (one more MFMA against an all-ones fragment), i.e. the denominator is the sum of exactly
// the quantised numerators.  Error of the log-linear mantissa + the rounding: 3.1 % rms per probability against 2.65 % for
// round-to-nearest e4m3 of the exact exponential (a constant factor of 1.045 cancels in the ratio); tools/probes/cvt_pk_u8.cpp.
// The running maximum is updated when a row's maximum grows by more than 2^A8_TAU; the bytes encode p 2^A8_OFF so that
// p = 2^A8_TAU is the top binade of e4m3 (byte 120 = 2^8) and p = 2^-12 its smallest normal; the B-side block scale 127 - A8_OFF
// of the two MFMAs undoes the shift.
constexpr float A8_TAU = 2.0f;
constexpr int A8_OFF = 6;
constexpr float A8_ADJ = 0.05f;                  // minimises the rms error of the log-linear map

// Measured and not adopted (8 x 4096 x 5 heads, 120 us as built here; tools/attn_tail.py, profiles/r05_attn_mx8_study.txt): two 64-key
// tiles per ring stage (half the workgroup barriers) 126 us; a software pipeline over the tiles (S of tile kt + 1 and the V^T reads
// of tile kt issued ahead of tile kt's softmax, scores double-buffered, ring of four) 159 us at 3 waves per SIMD (the Q fragment
// spills and its reload drains vmcnt) and 134 us at 2.  Knock-outs of that build: barrier + fragment reads + S MFMAs alone are half of
// the time — every wave reads the whole K and V tile out of LDS for its 32 queries; 64 queries per wave is the next step.
__global__ void __launch_bounds__(256, 3) attn_d64_mx8_kernel(const AttnMx8Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem8[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int hi = lane >> 5;
  const int unit = range_index(blockIdx.x, p.G);           // (XCD-aware unit order: see attn_d64_v2_kernel)
  if (unit >= p.units) return;
  const int bh = unit / p.qtiles;
  const int qt = unit - bh * p.qtiles;
  const int b = bh / p.heads;
  const int h = bh - b * p.heads;
  const long long row0 = (long long)b * p.n;
  const uint8_t* __restrict__ X = p.x8 + row0 * p.ld8;
  const int qi = qt * 128 + wave * 32 + l31;
  const bool qok = qi < p.n;
  // ---- Q fragment (the B operand of S^T = K Q^T) and its block scale: block 2 h + hi of the row
  i32x8_t qf;
  int qsc;
  {
    const long long qr = qok ? qi : 0;
    const uint8_t* g = X + qr * p.ld8 + h * 64;
    const u32x4 a = *reinterpret_cast<const u32x4*>(g + 16 * hi), bb = *reinterpret_cast<const u32x4*>(g + 32 + 16 * hi);
    i32x8_t t = {(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)bb[0], (int)bb[1], (int)bb[2], (int)bb[3]};
    qf = t;
    const int bq = 2 * h + hi;
    qsc = (int)(p.sc[(long long)(bq >> 2) * p.rows + row0 + qr] >> (8 * (bq & 3)));
  }
  // ---- staging: per wave and tile one K piece, one V piece (16 keys x 64 B each) and the tile's 64 scale dwords (every wave
  // writes the same 256 bytes: a uniform count of three loads per stage and wave).  The LDS-DMA writes lane-contiguous 16-byte
  // slots, so the bank swizzle is applied on the SOURCE side: LDS slot j of key row r holds head dims 16 (j ^ g(r)) .. + 15 with
  //   K: g(r) = ((r >> 2) & 1) | (((r >> 4) & 1) << 1)   (the 16 lanes of a ds_read_b128 phase read rows {0..3, 16..19, 4..7, 20..23} + c)
  //   V: g(r) = (r >> 2) & 1                              (the 16 lanes of a ds_read_b64_tr_b8 read rows r0 .. r0 + 7, 16 bytes each)
  const int kb0 = p.C / 32 + 2 * h;               // k's first block of this head: even -> both of its scales sit in one dword
  const unsigned kvbytes = (unsigned)((long long)(p.n - 1) * p.ld8 + 64);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(X + p.C + h * 64), 0, kvbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(X + 2 * p.C + h * 64), 0, kvbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.sc + (long long)(kb0 >> 2) * p.rows + row0), 0,
                                                                       (unsigned)(p.n * 4), 0x00020000);
  const unsigned row_voff = (unsigned)((long long)(wave * 16 + (lane >> 2)) * p.ld8);
  const unsigned k_voff = row_voff + (((lane & 3) ^ (((lane >> 4) & 1) | ((wave & 1) << 1))) << 4);
  const unsigned v_voff = row_voff + (((lane & 3) ^ ((lane >> 4) & 1)) << 4);
  const int tile_step = 64 * p.ld8;
  auto stage = [&](int st, int kt) __attribute__((always_inline)) {
    char* kbuf = smem8 + st * A8_STAGE;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (__attribute__((address_space(3))) void*)(kbuf + wave * 1024), 16, k_voff, kt * tile_step, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void*)(kbuf + A8_TILE + wave * 1024), 16, v_voff, kt * tile_step, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(kbuf + 2 * A8_TILE), 4, (unsigned)(lane * 4), kt * 256, 0, 0);
  };
  // ---- per-lane LDS offsets
  const int kperm = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);       // the key (of a 32-key half) this fragment lane holds
  const int kg = ((kperm >> 2) & 1) | (((kperm >> 4) & 1) << 1);
  const int k_off0 = kperm * 64 + ((hi ^ kg) << 4);                             // head dims 16 hi .. + 15 (scale block 0)
  const int k_off1 = kperm * 64 + (((2 + hi) ^ kg) << 4);                       // head dims 32 + 16 hi .. (scale block 1); + 2048: keys + 32
  const int s_off = 2 * A8_TILE + kperm * 4;
  const int s_shift = 8 * ((kb0 & 3) + hi);
  const int tr_i = lane & 15;
  // V^T: key row 16 hi + (i >> 1) (+ 8 g + 32 t), head dims 32 dt + 16 ((lane >> 4) & 1) + 8 (i & 1) .. + 7; (row >> 2) & 1 = i >> 3
  const int v_row = A8_TILE + (16 * hi + (tr_i >> 1)) * 64 + 8 * (tr_i & 1);
  const int v_off0 = v_row + (((0 + ((lane >> 4) & 1)) ^ (tr_i >> 3)) << 4);   // dt = 0
  const int v_off1 = v_row + (((2 + ((lane >> 4) & 1)) ^ (tr_i >> 3)) << 4);   // dt = 1; + 512 g + 2048 t

  f32x16 o_acc[2], l_acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) o_acc[0][r] = o_acc[1][r] = l_acc[r] = 0.f;
  i32x8_t ones = {0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838};   // e4m3 1.0
  asm volatile("" : "+v"(ones));                            // (kept in eight registers: left alone hipcc rebuilds the fragment every tile)
  float m_run = -INFINITY;
  const int ntiles = (p.n + 63) / 64;
  const float c = p.scale_log2e;
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  stage(0, 0);
  if (ntiles > 1) stage(1, 1);
  int st = 0;
  for (int kt = 0; kt < ntiles; ++kt) {
    if (kt + 1 < ntiles) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");       // the tile after this one stays in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 2 < ntiles) {
      int s2 = st + 2;
      if (s2 >= A8_NST) s2 -= A8_NST;
      stage(s2, kt + 2);
    }
    const char* buf = smem8 + st * A8_STAGE;
    f32x16 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const i32x8_t kf = lds_read_frag32(buf + t * 2048 + k_off0, buf + t * 2048 + k_off1);
      const int ksc = (int)(*reinterpret_cast<const uint32_t*>(buf + s_off + t * 128) >> s_shift);
      f32x16 z;
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] = 0.f;
      s[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf, z, 0, 0, 0, ksc, 0, qsc);
    }
    if (kt * 64 + 64 > p.n) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 64 + t * 32 + 16 * hi + r;              // (the permuted fragment rows: register r = key 16 hi + r)
          if (key >= p.n) s[t][r] = -INFINITY;
        }
    }
    float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s[0][r]), s[1][r]);     // (v_max3_f32)
    mx = half_swap_max(mx);
    const float mxs = mx * c;
    if (!__all(mxs - m_run <= A8_TAU)) {                    // wave-uniform: some query's maximum grew by more than 2^A8_TAU
      const float m_new = fmaxf(m_run, mxs);
      const float alpha = fast_exp2(m_run - m_new);
      m_run = m_new;
      l_acc[0] *= alpha;                                     // (every row of the ones product holds the same sums: row 0 is the one kept)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[i][r] *= alpha;
    }
    // ---- P as e4m3 bytes: the lane's own scores in register order (registers 0..3 = keys 16 hi + r of the first 32, 4..7 of the second)
    const float c8 = 8.0f * c, k8 = 8.0f * (A8_OFF + 7) + A8_ADJ - 8.0f * m_run;
    const f32x2v c2 = {c8, c8}, k2 = {k8, k8};
    i32x8_t pf;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const f32x2v e0 = __builtin_elementwise_fma(f32x2v{s[t][q4 * 4], s[t][q4 * 4 + 1]}, c2, k2);
        const f32x2v e1 = __builtin_elementwise_fma(f32x2v{s[t][q4 * 4 + 2], s[t][q4 * 4 + 3]}, c2, k2);
        unsigned w = __builtin_amdgcn_cvt_pk_u8_f32(e0[0], 0u, 0u);
        w = __builtin_amdgcn_cvt_pk_u8_f32(e0[1], 1u, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(e1[0], 2u, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(e1[1], 3u, w);
        pf[t * 4 + q4] = (int)w;
      }
    l_acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ones, pf, l_acc, 0, 0, 0, 127, 0, 127 - A8_OFF);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const unsigned vb = lds_offset(buf) + (dt ? v_off1 : v_off0);
      u32x2 w[4];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 2; ++g) w[t * 2 + g] = lds_tr8_b64(vb, t * 2048 + g * 512);
      lds_tr_wait(w[0], w[1], w[2], w[3]);
      const i32x8_t vf = {(int)w[0][0], (int)w[0][1], (int)w[1][0], (int)w[1][1], (int)w[2][0], (int)w[2][1], (int)w[3][0], (int)w[3][1]};
      o_acc[dt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, pf, o_acc[dt], 0, 0, 0, 127, 0, 127 - A8_OFF);
    }
    st = st + 1 == A8_NST ? 0 : st + 1;
  }
  const float inv = p.v_inv / l_acc[0];
  const long long m = row0 + qi;
  if (qok) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int d = dt * 32 + qd * 8 + hi * 4;
        u32x2 pk = {pack_bf16x2(o_acc[dt][qd * 4 + 0] * inv, o_acc[dt][qd * 4 + 1] * inv),
                    pack_bf16x2(o_acc[dt][qd * 4 + 2] * inv, o_acc[dt][qd * 4 + 3] * inv)};
        *reinterpret_cast<u32x2*>(p.o + m * p.ldo + h * 64 + d) = pk;
      }
  }
  if (p.q8_out) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = o_acc[dt][r] * inv;
      uint32_t q[4], sb;
      mx8_quant_acc16(v, q, sb);
      if (qok) {
        const int blk = h * 2 + dt;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) *reinterpret_cast<uint32_t*>(p.q8_out + m * p.ld_q8 + blk * 32 + qd * 8 + hi * 4) = q[qd];
        if (hi == 0) reinterpret_cast<uint8_t*>(p.q8_scale)[((long long)(blk >> 2) * p.rows + m) * 4 + (blk & 3)] = (uint8_t)sb;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// head_dim 512, one head (the AutoencoderKL mid-block attention, reference sgm/modules/diffusionmodules/model.py:236-260:
// xformers.ops.memory_efficient_attention on [B, H*W, 512]).  512 dims do not fit one wave's registers (O alone would be
// 256 accumulator registers per 32 queries), so the head dimension is SPLIT OVER THE FOUR WAVES of a workgroup:
//   * one workgroup = 32 queries; a K|V stage = 32 keys x 512 dims of each, kept as 8 + 8 sub-tiles of [32 keys][64 dims]
//     in the 128-byte-row swizzled image of the head_dim-64 kernel (so its fragment / transpose-read addressing carries over);
//   * wave w owns dims 128w .. 128w+127: it accumulates the PARTIAL scores S^T_w = K[:, dims_w] Q^T[dims_w, :] (8 MFMAs), the
//     four partial tiles are exchanged through LDS (fp32, 4 KiB per wave) and summed by every wave in the same order (so all
//     waves hold bit-identical scores, maxima and sums), each wave runs the online softmax redundantly (16 exponentials per
//     lane and tile) and then accumulates ITS 128 output dims O^T_w += V^T[dims_w, :] P^T (8 MFMAs);
//   * two stages of 64 KiB (the next K|V tile is in flight while one is consumed) + 16 KiB of exchange = 144 KiB, one
//     workgroup per CU.  No [N, N] score tensor exists anywhere.
template <int N> UDT_DEVINL void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct Attn512Params {
  const uint16_t* q;
  const uint16_t* k;
  const uint16_t* v;
  uint16_t* o;
  const uint16_t* zero;
  int nq, nk;
  int ldq, ldk, ldv, ldo;
  long long sq, sk, sv, so;
  float scale_log2e;
  // key split (udt_attn512_split_fwd): workgroup z of ksplit walks key tiles [z * kt_per, (z + 1) * kt_per) and parks its
  // UNNORMALISED partial result — fp32 O [b][z][nq][512] and (running maximum, row sum) [b][z][nq][2] — for the merge kernel
  int ksplit, kt_per;
  float* part_o;
  float* part_ml;
};
constexpr int A5_KEYS = 32;
constexpr int A5_SUB = A5_KEYS * 128;              // one [32 keys][64 dims] sub-tile
constexpr int A5_HALF = 8 * A5_SUB;                // K (or V) part of a stage: 32 KiB
constexpr int A5_STAGE = 2 * A5_HALF;
template <int QT> constexpr int a5_smem() { return 2 * A5_STAGE + 4 * QT * 4096; }

// QT: 32-query tiles per workgroup = groups of four waves (QT = 2: eight waves, two per SIMD — the second group halves the
// K|V bytes staged per FLOP and gives every SIMD a second wave to issue while the first waits on LDS / MFMA results; one
// wave per SIMD, which is all a 144 KiB workgroup of four waves allows, left every latency exposed: 274 -> 365 TFLOP/s with
// two query tiles per WAVE, see profiles/r03_attn512.txt).
// Staging order: K and V halves of a stage are refilled SEPARATELY as soon as they are free — K(t + 2) right after the score
// exchange barrier of tile t (every wave is past its K reads), V(t + 1) right after the top barrier of tile t (every wave is
// past P V of tile t - 1).
template <int QT>
UDT_DEVINL void attn_d512_body(const Attn512Params& p) {
  constexpr int NWV = 4 * QT;                              // waves per workgroup
  constexpr int PCS = 32 / NWV;                            // K (and V) pieces per wave and tile
  extern __shared__ __attribute__((aligned(16))) char smem5[];
  char* const xch = smem5 + 2 * A5_STAGE;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int hi = lane >> 5;
  const int qt = wave >> 2, dw = wave & 3;                 // query tile, dims quarter
  const int b = blockIdx.y;
  const uint16_t* __restrict__ Q = p.q + (long long)b * p.sq;
  const uint16_t* __restrict__ K = p.k + (long long)b * p.sk;
  const uint16_t* __restrict__ V = p.v + (long long)b * p.sv;
  uint16_t* __restrict__ O = p.o + (long long)b * p.so;
  const int qi = blockIdx.x * (32 * QT) + qt * 32 + l31;
  const bool qok = qi < p.nq;
  const int d0 = dw * 128;                                 // this wave's dims
  bf16x8_t qf[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const uint16_t* g = qok ? (Q + (long long)qi * p.ldq + d0 + ks * 16 + hi * 8) : p.zero;
    qf[ks] = *reinterpret_cast<const bf16x8_t*>(g);
  }
  // staging: 32 K pieces + 32 V pieces of 1 KiB (8 keys x 128 B of one sub-tile) per tile, PCS + PCS per wave
  const int l3 = lane >> 3, pslot = lane & 7;
  const unsigned kbytes = (unsigned)(((long long)(p.nk - 1) * p.ldk + 512) * 2);
  const unsigned vbytes = (unsigned)(((long long)(p.nk - 1) * p.ldv + 512) * 2);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(K), 0, kbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(V), 0, vbytes, 0x00020000);
  unsigned kvo[PCS], vvo[PCS];
#pragma unroll
  for (int i = 0; i < PCS; ++i) {
    const int piece = wave + NWV * i;                      // sub-tile piece >> 2, key rows (piece & 3) * 8 ..
    const int sub = piece >> 2, row = (piece & 3) * 8 + l3;
    const int doff = sub * 64 + (pslot ^ ((row >> 1) & 7)) * 8;
    kvo[i] = (unsigned)(((long long)row * p.ldk + doff) * 2);
    vvo[i] = (unsigned)(((long long)row * p.ldv + doff) * 2);
  }
  const int kstep = A5_KEYS * p.ldk * 2, vstep = A5_KEYS * p.ldv * 2;          // bytes per 32-key tile
  auto stage_k = [&](int kt) {
    char* kbuf = smem5 + (kt & 1) * A5_STAGE;
#pragma unroll
    for (int i = 0; i < PCS; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (__attribute__((address_space(3))) void*)(kbuf + (wave + NWV * i) * 1024), 16, kvo[i], kt * kstep, 0, 0);
  };
  auto stage_v = [&](int kt) {
    char* vbuf = smem5 + (kt & 1) * A5_STAGE + A5_HALF;
#pragma unroll
    for (int i = 0; i < PCS; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void*)(vbuf + (wave + NWV * i) * 1024), 16, vvo[i], kt * vstep, 0, 0);
  };
  // per-lane LDS offsets inside a sub-tile (as attn_d64_v2_kernel)
  const int swz = (l31 >> 1) & 7;
  int koff_l[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff_l[ks] = l31 * 128 + (((ks * 2 + hi) ^ swz) << 4);
  const int tr_i = lane & 15, tr_g1 = (lane >> 4) & 1;
  const int tr_row = 4 * hi + (tr_i >> 2);
  const int tr_swz = (tr_row >> 1) & 7;
  const int tr_chunk = 2 * tr_g1 + ((tr_i & 3) >> 1);
  const int tr_base = tr_row * 128 + (tr_i & 1) * 8;
  int voff_a[2], voff_b[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    const int c = dt * 4 + tr_chunk;
    voff_a[dt] = tr_base + ((c ^ tr_swz) << 4);
    voff_b[dt] = tr_base + 1024 + ((c ^ tr_swz ^ 4) << 4);
  }
  const int sub0 = dw * 2;                                 // this wave's two sub-tiles (128 dims)
  char* const xq = xch + qt * (4 * 4096);                  // exchange area of this query tile's four waves

  f32x16 o_acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  // this workgroup's key tiles: all of them, or slice blockIdx.z of the key split
  const int kt_begin = p.ksplit > 1 ? (int)blockIdx.z * p.kt_per : 0;
  int ntiles = (p.nk + A5_KEYS - 1) / A5_KEYS;
  if (p.ksplit > 1 && kt_begin + p.kt_per < ntiles) ntiles = kt_begin + p.kt_per;
  const float c = p.scale_log2e;
  // issue order: K(0) V(0) K(1) | tile t: [top barrier] V(t+1) ... [exchange barrier] K(t+2)
  stage_k(kt_begin);
  stage_v(kt_begin);
  if (kt_begin + 1 < ntiles) stage_k(kt_begin + 1);
  for (int kt = kt_begin; kt < ntiles; ++kt) {
    const int st = kt & 1;
    // K(kt) landed: the loads behind it are V(kt) and K(kt+1) (PCS + PCS per wave) while both exist
    if (kt + 2 < ntiles) wait_vm<2 * PCS>(); else wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // every wave is past P V of tile kt - 1: V slot (kt+1)&1 and xch are free
    asm volatile("" ::: "memory");
    if (kt + 1 < ntiles) stage_v(kt + 1);
    const char* kbuf = smem5 + st * A5_STAGE;
    const char* vbuf = kbuf + A5_HALF;
    // partial scores over this wave's 128 dims: two accumulators (one per sub-tile) keep the MFMA chain from serialising
    f32x16 sp0, sp1;
#pragma unroll
    for (int r = 0; r < 16; ++r) sp0[r] = 0.f, sp1[r] = 0.f;
    {
      bf16x8_t kf0[4], kf1[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        kf0[ks] = lds_read_frag(kbuf + sub0 * A5_SUB + koff_l[ks]);
        kf1[ks] = lds_read_frag(kbuf + (sub0 + 1) * A5_SUB + koff_l[ks]);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        sp0 = mfma32(kf0[ks], qf[ks], sp0);
        sp1 = mfma32(kf1[ks], qf[4 + ks], sp1);
      }
    }
    // exchange: xq[dw][q4][lane] (16 B), then every wave of the query tile sums the four partial tiles in the same order
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      f32x4 v = {sp0[q4 * 4 + 0] + sp1[q4 * 4 + 0], sp0[q4 * 4 + 1] + sp1[q4 * 4 + 1], sp0[q4 * 4 + 2] + sp1[q4 * 4 + 2],
                 sp0[q4 * 4 + 3] + sp1[q4 * 4 + 3]};
      *reinterpret_cast<f32x4*>(xq + dw * 4096 + q4 * 1024 + lane * 16) = v;
    }
    // V(kt) landed too (behind it: K(kt+1), V(kt+1))
    if (kt + 2 < ntiles) wait_vm<2 * PCS>(); else wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // every wave is past its K reads of tile kt: K slot kt&1 is free
    asm volatile("" ::: "memory");
    if (kt + 2 < ntiles) stage_k(kt + 2);
    f32x16 s;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      f32x4 a = *reinterpret_cast<const f32x4*>(xq + q4 * 1024 + lane * 16);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(xq + w * 4096 + q4 * 1024 + lane * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += t[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) s[q4 * 4 + e] = a[e];
    }
    if (kt * A5_KEYS + A5_KEYS > p.nk) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * A5_KEYS + 8 * (r >> 2) + 4 * hi + (r & 3);
        if (key >= p.nk) s[r] = -INFINITY;
      }
    }
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = half_swap_max(mx);
    const float m_new = fmaxf(m_run, mx * c);
    if (!__all(m_new == m_run)) {                          // (wave-uniform: skip the 64 multiplies when no maximum moved)
      const float alpha = fast_exp2(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[i][r] *= alpha;
    }
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = fast_exp2(s[r] * c - m_run);
      s[r] = pv;
      psum += pv;
    }
    l_run += psum;
#pragma unroll
    for (int s4 = 0; s4 < 2; ++s4) {                       // two 16-key steps
      u32x4 pk;
      pk[0] = pack_bf16x2(s[s4 * 8 + 0], s[s4 * 8 + 1]);
      pk[1] = pack_bf16x2(s[s4 * 8 + 2], s[s4 * 8 + 3]);
      pk[2] = pack_bf16x2(s[s4 * 8 + 4], s[s4 * 8 + 5]);
      pk[3] = pack_bf16x2(s[s4 * 8 + 6], s[s4 * 8 + 7]);
      const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pk);
      u32x4 vv[4];
      u32x2 tl[4], th[4];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const unsigned sb = lds_offset(vbuf) + (sub0 + j) * A5_SUB;
          u32x2 lo = lds_tr16_b64(sb + voff_a[dt], s4 * 2048), hh = lds_tr16_b64(sb + voff_b[dt], s4 * 2048);
          tl[j * 2 + dt] = lo;
          th[j * 2 + dt] = hh;
        }
      lds_tr_wait(tl[0], th[0], tl[1], th[1]);
      lds_tr_wait(tl[2], th[2], tl[3], th[3]);
#pragma unroll
      for (int t = 0; t < 4; ++t) vv[t] = u32x4{tl[t][0], tl[t][1], th[t][0], th[t][1]};
#pragma unroll
      for (int t = 0; t < 4; ++t) o_acc[t] = mfma32(__builtin_bit_cast(bf16x8_t, vv[t]), pf, o_acc[t]);
    }
  }
  const float l_tot = half_swap_sum(l_run);
  if (p.ksplit > 1) {
    // park the slice: unnormalised O (fp32, this wave's 128 dims of its 32 queries) and, once per query, (m, l)
    if (qok) {
      const long long row = ((long long)b * p.ksplit + blockIdx.z) * p.nq + qi;
      float* po = p.part_o + row * 512;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int d = d0 + t * 32 + qd * 8 + hi * 4;
          f32x4 v = {o_acc[t][qd * 4 + 0], o_acc[t][qd * 4 + 1], o_acc[t][qd * 4 + 2], o_acc[t][qd * 4 + 3]};
          *reinterpret_cast<f32x4*>(po + d) = v;
        }
      if (dw == 0 && hi == 0) {
        f32x2 ml = {m_run, l_tot};
        *reinterpret_cast<f32x2*>(p.part_ml + row * 2) = ml;
      }
    }
    return;
  }
  const float inv = 1.0f / l_tot;
  if (qok) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int d = d0 + t * 32 + qd * 8 + hi * 4;
        u32x2 pk = {pack_bf16x2(o_acc[t][qd * 4 + 0] * inv, o_acc[t][qd * 4 + 1] * inv),
                    pack_bf16x2(o_acc[t][qd * 4 + 2] * inv, o_acc[t][qd * 4 + 3] * inv)};
        *reinterpret_cast<u32x2*>(O + (long long)qi * p.ldo + d) = pk;
      }
  }
}

// merge of the key split: per query, O = sum_z 2^(m_z - m) O_z / sum_z 2^(m_z - m) l_z with m = max_z m_z (the slices carry
// their running maxima in the log2 domain); 128 threads = one query, four dims each; slices in index order -> deterministic
__global__ void __launch_bounds__(128) attn_d512_merge_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                              uint16_t* __restrict__ o, int nq, int ksplit, int ldo, long long so) {
  const int qi = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  float m = -INFINITY;
  for (int z = 0; z < ksplit; ++z) m = fmaxf(m, part_ml[(((long long)b * ksplit + z) * nq + qi) * 2]);
  float l = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < ksplit; ++z) {
    const long long row = ((long long)b * ksplit + z) * nq + qi;
    const f32x2 ml = *reinterpret_cast<const f32x2*>(part_ml + row * 2);
    const float w = fast_exp2(ml[0] - m);
    l += w * ml[1];
    const f32x4 v = *reinterpret_cast<const f32x4*>(part_o + row * 512 + t * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += w * v[e];
  }
  const float inv = 1.0f / l;
  u32x2 pk = {pack_bf16x2(acc[0] * inv, acc[1] * inv), pack_bf16x2(acc[2] * inv, acc[3] * inv)};
  *reinterpret_cast<u32x2*>(o + (long long)b * so + (long long)qi * ldo + t * 4) = pk;
}

// (two plain kernels around the template body: a launch bound that depends on a template parameter left the host stub of an
//  anonymous-namespace kernel template undefined at link time)
__global__ void __launch_bounds__(256, 1) attn_d512_q32_kernel(const Attn512Params p) { attn_d512_body<1>(p); }
__global__ void __launch_bounds__(512, 1) attn_d512_q64_kernel(const Attn512Params p) { attn_d512_body<2>(p); }

// ---------------------------------------------------------------------------------------------------
// short-context attention: L <= 16 keys, head_dim a multiple of 64
struct XattnParams {
  const uint16_t* q;
  const uint16_t* k;
  const uint16_t* v;
  uint16_t* o;
  float* probs;
  int heads, head_dim, nq, L;
  int ldq, ldkv, ldo;
  float scale;
};

constexpr int XL_MAX = 16;

// grid: (ceil(nq/256), heads, batch); each lane owns one query of head blockIdx.y
__global__ void __launch_bounds__(256) xattn_kernel(const XattnParams p) {
  extern __shared__ __attribute__((aligned(16))) char xsm[];
  float* ks = reinterpret_cast<float*>(xsm);                 // [L][head_dim] fp32
  float* vs = ks + p.L * p.head_dim;                         // [L][head_dim] fp32
  const int h = blockIdx.y;
  const int b = blockIdx.z;
  const int D = p.head_dim;
  // stage K and V of (b, h) as fp32
  for (int i = threadIdx.x; i < p.L * D; i += 256) {
    const int l = i / D;
    const int d = i - l * D;
    const long long off = ((long long)b * p.L + l) * p.ldkv + h * D + d;
    ks[i] = bf16_bits_to_f32(p.k[off]);
    vs[i] = bf16_bits_to_f32(p.v[off]);
  }
  __syncthreads();
  const int qi = blockIdx.x * 256 + threadIdx.x;
  if (qi >= p.nq) return;
  const uint16_t* qrow = p.q + ((long long)b * p.nq + qi) * p.ldq + h * D;
  float sc[XL_MAX];
#pragma unroll
  for (int l = 0; l < XL_MAX; ++l) sc[l] = 0.f;
  for (int d0 = 0; d0 < D; d0 += 8) {
    const u32x4 qv = *reinterpret_cast<const u32x4*>(qrow + d0);
    float qf[8] = {bf16_lo(qv[0]), bf16_hi(qv[0]), bf16_lo(qv[1]), bf16_hi(qv[1]),
                   bf16_lo(qv[2]), bf16_hi(qv[2]), bf16_lo(qv[3]), bf16_hi(qv[3])};
#pragma unroll
    for (int l = 0; l < XL_MAX; ++l) {
      if (l < p.L) {
        const float* kr = ks + l * D + d0;
#pragma unroll
        for (int j = 0; j < 8; ++j) sc[l] += qf[j] * kr[j];
      }
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int l = 0; l < XL_MAX; ++l)
    if (l < p.L) {
      sc[l] *= p.scale;
      mx = fmaxf(mx, sc[l]);
    }
  if (p.L == 1) {
    // a single context token: the reference applies a sigmoid instead of a softmax (attention.py:159-162)
    sc[0] = 1.0f / (1.0f + __expf(-sc[0]));
  } else {
    float sum = 0.f;
#pragma unroll
    for (int l = 0; l < XL_MAX; ++l)
      if (l < p.L) {
        sc[l] = __expf(sc[l] - mx);
        sum += sc[l];
      }
    const float inv = 1.0f / sum;
#pragma unroll
    for (int l = 0; l < XL_MAX; ++l) sc[l] = (l < p.L) ? sc[l] * inv : 0.f;
  }
  if (p.probs) {
    float* pr = p.probs + (((long long)b * p.heads + h) * p.nq + qi) * p.L;
#pragma unroll
    for (int l = 0; l < XL_MAX; ++l)
      if (l < p.L) pr[l] = sc[l];
  }
  uint16_t* orow = p.o + ((long long)b * p.nq + qi) * p.ldo + h * D;
  for (int d0 = 0; d0 < D; d0 += 8) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int l = 0; l < XL_MAX; ++l) {
      if (l < p.L) {
        const float* vr = vs + l * D + d0;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += sc[l] * vr[j];
      }
    }
    u32x4 pk = {pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]),
                pack_bf16x2(acc[6], acc[7])};
    *reinterpret_cast<u32x4*>(orow + d0) = pk;
  }
}

// ---------------------------------------------------------------------------------------------------
// in-place row softmax over bf16 [rows, cols]; one workgroup per row, fp32 math
__global__ void __launch_bounds__(256) softmax_rows_kernel(uint16_t* x, int cols, int ld) {
  __shared__ float red[8];
  uint16_t* row = x + (long long)blockIdx.x * ld;
  const int tid = threadIdx.x;
  const int nch = cols >> 3;
  float mx = -INFINITY;
  for (int ch = tid; ch < nch; ch += 256) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(row + ch * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) mx = fmaxf(mx, fmaxf(bf16_lo(v[j]), bf16_hi(v[j])));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int ch = tid; ch < nch; ch += 256) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(row + ch * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += __expf(bf16_lo(v[j]) - mx) + __expf(bf16_hi(v[j]) - mx);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
  __syncthreads();
  sum = red[4] + red[5] + red[6] + red[7];
  const float inv = 1.0f / sum;
  for (int ch = tid; ch < nch; ch += 256) {
    u32x4 v = *reinterpret_cast<const u32x4*>(row + ch * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      v[j] = pack_bf16x2(__expf(bf16_lo(v[j]) - mx) * inv, __expf(bf16_hi(v[j]) - mx) * inv);
    *reinterpret_cast<u32x4*>(row + ch * 8) = v;
  }
}

}  // namespace

static int attn_fwd_impl(bool vrow, const void* q, const void* k, const void* vt, void* o, int32_t batch, int32_t heads,
                         int32_t nq, int32_t nk, int32_t ldq, int32_t ldk, int32_t ldvt, int32_t ldo,
                         int64_t q_bstride, int64_t k_bstride, int64_t vt_bstride, int64_t o_bstride,
                         float scale, void* stream, void* q8_out = nullptr, void* q8_scale = nullptr, int32_t ld_q8 = 0) {
  if (!q || !k || !vt || !o) return UDT_ERR_BAD_ARG;
  if (q8_out && (!vrow || !q8_scale || ld_q8 < heads * 64 || ld_q8 % 4 != 0 || heads % 2 != 0)) return UDT_ERR_BAD_ARG;
  if (batch <= 0 || heads <= 0 || nq <= 0 || nk <= 0) return UDT_ERR_BAD_SHAPE;
  // (V^T operand: its rows are the keys — 16-byte row pieces need nk % 8 == 0.  Row-major V: any nk — keys past nk read zeros through
  //  the buffer descriptor and are masked; the 2 x 2 middle block of a 128 x 128 image has 4 keys)
  if ((!vrow && nk % 8 != 0) || ldq % 8 != 0 || ldk % 8 != 0 || ldvt % 8 != 0 || ldo % 4 != 0) return UDT_ERR_BAD_SHAPE;
  AttnParams p;
  p.q = reinterpret_cast<const uint16_t*>(q);
  p.k = reinterpret_cast<const uint16_t*>(k);
  p.vt = reinterpret_cast<const uint16_t*>(vt);
  p.o = reinterpret_cast<uint16_t*>(o);
  p.zero = udt_zero_page();
  if (!p.zero) return UDT_ERR_HIP;
  p.heads = heads; p.nq = nq; p.nk = nk;
  p.ldq = ldq; p.ldk = ldk; p.ldvt = ldvt; p.ldo = ldo;
  p.sq = q_bstride; p.sk = k_bstride; p.svt = vt_bstride; p.so = o_bstride;
  p.scale_log2e = scale * 1.4426950408889634f;
  p.q8_out = reinterpret_cast<uint8_t*>(q8_out); p.q8_scale = reinterpret_cast<uint32_t*>(q8_scale); p.ld_q8 = ld_q8;
  p.q8_rows = (long long)batch * nq;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  UdtProfScope prof(2, s);
  if (prof.rec) {
    char tag[96];
    snprintf(tag, sizeof(tag), "attn%s B=%d H=%d nq=%d nk=%d", q8_out ? "+q8" : "", batch, heads, nq, nk);
    udt_prof_tag(prof.rec, tag);
  }
  dim3 grid((nq + 127) / 128, batch * heads);
  if (vrow) {
    // row-major V: the three-stage kernel; it addresses K / V through 31-bit buffer offsets
    if ((long long)nk * ldk * 2 >= (1LL << 31) || (long long)nk * ldvt * 2 >= (1LL << 31)) return UDT_ERR_BAD_SHAPE;
    static std::atomic<int> attr_done{0};
    constexpr int smem = A2_SMEM;
    if (!attr_done.load()) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_d64_v2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      if (e != hipSuccess) return udt_set_hip_error(e);
      attr_done.store(1);
    }
    p.qtiles = (nq + 127) / 128;
    p.units = p.qtiles * batch * heads;
    p.G = (p.units + 7) / 8 * 8;
    hipLaunchKernelGGL(attn_d64_v2_kernel, dim3(p.G), dim3(256), smem, s, p);
  } else {
    hipLaunchKernelGGL(attn_d64_kernel<false>, grid, dim3(256), 0, s, p);      // V^T operand (udt_attn_fwd)
  }
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_attn_fwd(const void* q, const void* k, const void* vt, void* o, int32_t batch, int32_t heads,
                            int32_t nq, int32_t nk, int32_t ldq, int32_t ldk, int32_t ldvt, int32_t ldo,
                            int64_t q_bstride, int64_t k_bstride, int64_t vt_bstride, int64_t o_bstride,
                            float scale, void* stream) {
  return attn_fwd_impl(false, q, k, vt, o, batch, heads, nq, nk, ldq, ldk, ldvt, ldo, q_bstride, k_bstride, vt_bstride,
                       o_bstride, scale, stream);
}

extern "C" int udt_attn_rowv_fwd(const void* q, const void* k, const void* v, void* o, int32_t batch, int32_t heads,
                                 int32_t nq, int32_t nk, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                                 int64_t q_bstride, int64_t k_bstride, int64_t v_bstride, int64_t o_bstride,
                                 float scale, void* stream) {
  return attn_fwd_impl(true, q, k, v, o, batch, heads, nq, nk, ldq, ldk, ldv, ldo, q_bstride, k_bstride, v_bstride,
                       o_bstride, scale, stream);
}

extern "C" int udt_attn_rowv_q8_fwd(const void* q, const void* k, const void* v, void* o, int32_t batch, int32_t heads,
                                    int32_t nq, int32_t nk, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                                    int64_t q_bstride, int64_t k_bstride, int64_t v_bstride, int64_t o_bstride,
                                    float scale, void* q8_out, void* q8_scale, int32_t ld_q8, void* stream) {
  if (!q8_out) return UDT_ERR_BAD_ARG;
  // (the MX8 twin is addressed as rows b * nq + i: the bf16 output must be that matrix too)
  if (o_bstride != (int64_t)nq * ldo) return UDT_ERR_BAD_SHAPE;
  return attn_fwd_impl(true, q, k, v, o, batch, heads, nq, nk, ldq, ldk, ldv, ldo, q_bstride, k_bstride, v_bstride,
                       o_bstride, scale, stream, q8_out, q8_scale, ld_q8);
}

extern "C" int udt_attn_mx8_fwd(const void* qkv8, const void* qkv_scale, void* o, int32_t batch, int32_t heads, int32_t n,
                                int32_t ld8, int32_t ldo, float scale, float v_inv, void* q8_out, void* q8_scale, int32_t ld_q8,
                                void* stream) {
  if (!qkv8 || !qkv_scale || !o) return UDT_ERR_BAD_ARG;
  if (batch <= 0 || heads <= 0 || n <= 0 || n % 4 != 0) return UDT_ERR_BAD_SHAPE;
  const int C = heads * 64;
  // (16-byte fragment loads and LDS-DMA pieces; k's two block scales of a head share a dword when C / 32 is even; 31-bit offsets)
  if (ld8 < 3 * C || ld8 % 16 != 0 || ldo % 4 != 0 || (C / 32) % 2 != 0 || (long long)n * ld8 >= (1LL << 31)) return UDT_ERR_BAD_SHAPE;
  if ((reinterpret_cast<uintptr_t>(qkv8) & 15) || !(v_inv > 0.f)) return UDT_ERR_BAD_ARG;
  if (q8_out && (!q8_scale || ld_q8 < C || ld_q8 % 4 != 0)) return UDT_ERR_BAD_ARG;
  AttnMx8Params p;
  p.x8 = reinterpret_cast<const uint8_t*>(qkv8); p.sc = reinterpret_cast<const uint32_t*>(qkv_scale);
  p.o = reinterpret_cast<uint16_t*>(o);
  p.heads = heads; p.n = n; p.ld8 = ld8; p.ldo = ldo; p.C = C; p.rows = (long long)batch * n;
  p.scale_log2e = scale * 1.4426950408889634f; p.v_inv = v_inv;
  p.q8_out = reinterpret_cast<uint8_t*>(q8_out); p.q8_scale = reinterpret_cast<uint32_t*>(q8_scale); p.ld_q8 = ld_q8;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  UdtProfScope prof(2, s);
  if (prof.rec) {
    char tag[96];
    snprintf(tag, sizeof(tag), "attn-mx8%s B=%d H=%d nq=%d nk=%d", q8_out ? "+q8" : "", batch, heads, n, n);
    udt_prof_tag(prof.rec, tag);
  }
  p.qtiles = (n + 127) / 128;
  p.units = p.qtiles * batch * heads;
  p.G = (p.units + 7) / 8 * 8;
  hipLaunchKernelGGL(attn_d64_mx8_kernel, dim3(p.G), dim3(256), A8_NST * A8_STAGE, s, p);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

// key split plan: when the query tiles alone leave most CUs idle (a single image: 64 workgroups, each walking all 4096 keys
// alone), the keys are cut into up to 8 slices of >= 8 tiles (256 keys).  The slice count minimises the length of the launch in
// units of one unsplit workgroup — rounds of 256 workgroups (one per CU) x 1/slices — plus 2 % per slice for the parked partial
// results and the merge (64 query tiles: 4 slices, one round of a quarter; 144 tiles (768 x 768): 5 slices, three rounds of a fifth)
static int a5_key_split(int batch, int nq, int nk) {
  const long long wgs = (long long)((nq + 63) / 64) * batch;
  if (wgs >= 192) return 1;
  const int ntiles = (nk + A5_KEYS - 1) / A5_KEYS;
  int best = 1;
  double best_cost = 1.0;
  for (int ks = 2; ks <= 8 && ks * 8 <= ntiles; ++ks) {
    const double cost = (double)((wgs * ks + 255) / 256) / ks + 0.02 * ks;
    if (cost < best_cost - 1e-9) best_cost = cost, best = ks;
  }
  return best;
}

extern "C" size_t udt_attn512_workspace_bytes(int32_t batch, int32_t nq, int32_t nk) {
  if (batch <= 0 || nq <= 0 || nk <= 0) return 0;
  const int ks = a5_key_split(batch, nq, nk);
  return ks > 1 ? (size_t)batch * ks * nq * (512 + 2) * sizeof(float) : 0;
}

static int attn512_impl(const void* q, const void* k, const void* v, void* o, int32_t batch, int32_t nq, int32_t nk,
                        int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                        int64_t q_bstride, int64_t k_bstride, int64_t v_bstride, int64_t o_bstride,
                        float scale, void* workspace, size_t workspace_bytes, void* stream) {
  if (!q || !k || !v || !o) return UDT_ERR_BAD_ARG;
  if (batch <= 0 || nq <= 0 || nk <= 0) return UDT_ERR_BAD_SHAPE;
  if (ldq % 8 != 0 || ldk % 8 != 0 || ldv % 8 != 0 || ldo % 4 != 0 || ldq < 512 || ldk < 512 || ldv < 512 || ldo < 512) return UDT_ERR_BAD_SHAPE;
  if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v)) & 15) return UDT_ERR_BAD_SHAPE;
  if (reinterpret_cast<uintptr_t>(o) & 7) return UDT_ERR_BAD_SHAPE;
  if ((long long)nk * ldk * 2 >= (1LL << 31) || (long long)nk * ldv * 2 >= (1LL << 31)) return UDT_ERR_BAD_SHAPE;
  if (batch > 65535) return UDT_ERR_BAD_SHAPE;
  Attn512Params p;
  p.q = reinterpret_cast<const uint16_t*>(q);
  p.k = reinterpret_cast<const uint16_t*>(k);
  p.v = reinterpret_cast<const uint16_t*>(v);
  p.o = reinterpret_cast<uint16_t*>(o);
  p.zero = udt_zero_page();
  if (!p.zero) return UDT_ERR_HIP;
  p.nq = nq; p.nk = nk;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.sq = q_bstride; p.sk = k_bstride; p.sv = v_bstride; p.so = o_bstride;
  p.scale_log2e = scale * 1.4426950408889634f;
  p.ksplit = 1; p.kt_per = 0; p.part_o = nullptr; p.part_ml = nullptr;
  if (workspace) {
    const int ks = a5_key_split(batch, nq, nk);
    if (ks > 1) {
      const size_t need = (size_t)batch * ks * nq * (512 + 2) * sizeof(float);
      if (workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15)) return UDT_ERR_WORKSPACE;
      const int ntiles_all = (nk + A5_KEYS - 1) / A5_KEYS;
      p.kt_per = (ntiles_all + ks - 1) / ks;
      p.ksplit = (ntiles_all + p.kt_per - 1) / p.kt_per;       // (<= ks: every slice owns at least one tile)
      p.part_o = reinterpret_cast<float*>(workspace);
      p.part_ml = p.part_o + (size_t)batch * p.ksplit * nq * 512;
      if (p.ksplit < 2) p.ksplit = 1;
    }
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  UdtProfScope prof(2, s);
  if (prof.rec) {
    char tag[96];
    snprintf(tag, sizeof(tag), "attn512 B=%d nq=%d nk=%d ksplit=%d", batch, nq, nk, p.ksplit);
    udt_prof_tag(prof.rec, tag);
  }
  // 64 queries (eight waves) per workgroup when that still gives every CU work; 32 otherwise
  const bool two = (long long)((nq + 63) / 64) * batch * p.ksplit >= 224;
  static std::atomic<int> attr_done{0};
  if (!attr_done.load()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_d512_q32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, a5_smem<1>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_d512_q64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, a5_smem<2>());
    if (e != hipSuccess) return udt_set_hip_error(e);
    attr_done.store(1);
  }
  if (two) hipLaunchKernelGGL(attn_d512_q64_kernel, dim3((nq + 63) / 64, batch, p.ksplit), dim3(512), a5_smem<2>(), s, p);
  else hipLaunchKernelGGL(attn_d512_q32_kernel, dim3((nq + 31) / 32, batch, p.ksplit), dim3(256), a5_smem<1>(), s, p);
  UDT_CHECK_LAUNCH();
  if (p.ksplit > 1) {
    hipLaunchKernelGGL(attn_d512_merge_kernel, dim3(nq, batch), dim3(128), 0, s, p.part_o, p.part_ml, p.o, nq, p.ksplit, ldo,
                       (long long)o_bstride);
    UDT_CHECK_LAUNCH();
  }
  return UDT_OK;
}

extern "C" int udt_attn512_fwd(const void* q, const void* k, const void* v, void* o, int32_t batch, int32_t nq, int32_t nk,
                               int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                               int64_t q_bstride, int64_t k_bstride, int64_t v_bstride, int64_t o_bstride,
                               float scale, void* stream) {
  return attn512_impl(q, k, v, o, batch, nq, nk, ldq, ldk, ldv, ldo, q_bstride, k_bstride, v_bstride, o_bstride, scale, nullptr, 0,
                      stream);
}

extern "C" int udt_attn512_split_fwd(const void* q, const void* k, const void* v, void* o, int32_t batch, int32_t nq, int32_t nk,
                                     int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                                     int64_t q_bstride, int64_t k_bstride, int64_t v_bstride, int64_t o_bstride,
                                     float scale, void* workspace, size_t workspace_bytes, void* stream) {
  return attn512_impl(q, k, v, o, batch, nq, nk, ldq, ldk, ldv, ldo, q_bstride, k_bstride, v_bstride, o_bstride, scale, workspace,
                      workspace_bytes, stream);
}

extern "C" int udt_xattn_fwd(const void* q, const void* k, const void* v, void* o, float* probs, int32_t batch,
                             int32_t heads, int32_t head_dim, int32_t nq, int32_t L, int32_t ldq, int32_t ldkv,
                             int32_t ldo, float scale, void* stream) {
  if (!q || !k || !v || !o) return UDT_ERR_BAD_ARG;
  if (batch <= 0 || heads <= 0 || nq <= 0 || L <= 0 || L > XL_MAX) return UDT_ERR_BAD_SHAPE;
  if (head_dim <= 0 || head_dim % 64 != 0 || ldq % 8 != 0 || ldo % 8 != 0) return UDT_ERR_BAD_SHAPE;
  XattnParams p;
  p.q = reinterpret_cast<const uint16_t*>(q);
  p.k = reinterpret_cast<const uint16_t*>(k);
  p.v = reinterpret_cast<const uint16_t*>(v);
  p.o = reinterpret_cast<uint16_t*>(o);
  p.probs = probs;
  p.heads = heads; p.head_dim = head_dim; p.nq = nq; p.L = L;
  p.ldq = ldq; p.ldkv = ldkv; p.ldo = ldo;
  p.scale = scale;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  UdtProfScope prof(3, s);
  const size_t smem = (size_t)2 * L * head_dim * sizeof(float);
  dim3 grid((nq + 255) / 256, heads, batch);
  hipLaunchKernelGGL(xattn_kernel, grid, dim3(256), smem, s, p);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

// ---------------------------------------------------------------------------------------------------
// masked small attention (OCR scorer decoder): one workgroup per (head, sample); K and V of the head staged as fp32
// in LDS (row stride D + 1: lanes walk the keys), one wave per query: lanes own keys for the scores / softmax, then
// channels for the output
struct MattnParams {
  const uint16_t* q;
  const uint16_t* k;
  const uint16_t* v;
  uint16_t* o;
  const float* mask;
  const uint8_t* kpm;
  int heads, D, nq, lk;
  int ldq, ldk, ldv, ldo, ldmask;
  long long sq, sk, sv, so;
  float scale;
};

__global__ void __launch_bounds__(256) mattn_kernel(const MattnParams p) {
  extern __shared__ __attribute__((aligned(16))) char msm[];
  const int D = p.D, RS = D + 1;
  float* ks = reinterpret_cast<float*>(msm);                 // [lk][D + 1]
  float* vs = ks + p.lk * RS;                                // [lk][D + 1]
  float* qs = vs + p.lk * RS;                                // [4][64]
  float* ps = qs + 4 * 64;                                   // [4][256]
  const int h = blockIdx.x, b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < p.lk * D; i += 256) {
    const int l = i / D, d = i - l * D;
    ks[l * RS + d] = bf16_bits_to_f32(p.k[(long long)b * p.sk + (long long)l * p.ldk + h * D + d]);
    vs[l * RS + d] = bf16_bits_to_f32(p.v[(long long)b * p.sv + (long long)l * p.ldv + h * D + d]);
  }
  __syncthreads();
  const int rounds = (p.nq + 3) >> 2;
  for (int it = 0; it < rounds; ++it) {
    const int qi = it * 4 + wave;
    const bool active = qi < p.nq;
    if (active && lane < D) qs[wave * 64 + lane] = bf16_bits_to_f32(p.q[(long long)b * p.sq + (long long)qi * p.ldq + h * D + lane]);
    __syncthreads();
    float sc[4];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int l = lane + j * 64;
      float s = -INFINITY;
      if (active && l < p.lk) {
        s = 0.f;
        for (int d = 0; d < D; ++d) s += qs[wave * 64 + d] * ks[l * RS + d];
        s *= p.scale;
        if (p.mask) s += p.mask[(long long)qi * p.ldmask + l];
        if (p.kpm && p.kpm[(long long)b * p.lk + l]) s = -INFINITY;
      }
      sc[j] = s;
      mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sc[j] = (mx == -INFINITY || sc[j] == -INFINITY) ? 0.f : __expf(sc[j] - mx);
      sum += sc[j];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int l = lane + j * 64;
      if (l < p.lk) ps[wave * 256 + l] = sc[j] * inv;
    }
    __syncthreads();
    if (active && lane < D) {
      float acc = 0.f;
      for (int l = 0; l < p.lk; ++l) acc += ps[wave * 256 + l] * vs[l * RS + lane];
      p.o[(long long)b * p.so + (long long)qi * p.ldo + h * D + lane] = (uint16_t)(pack_bf16x2(acc, 0.f) & 0xffffu);
    }
    __syncthreads();
  }
}

extern "C" int udt_mattn_fwd(const void* q, const void* k, const void* v, void* o, const float* mask, const uint8_t* kpm,
                             int32_t batch, int32_t heads, int32_t head_dim, int32_t nq, int32_t lk, int32_t ldq,
                             int32_t ldk, int32_t ldv, int32_t ldo, int32_t ldmask, int64_t q_bstride, int64_t k_bstride,
                             int64_t v_bstride, int64_t o_bstride, float scale, void* stream) {
  if (!q || !k || !v || !o) return UDT_ERR_BAD_ARG;
  if (batch <= 0 || heads <= 0 || nq <= 0 || lk <= 0 || lk > 256 || batch > 65535) return UDT_ERR_BAD_SHAPE;
  if (head_dim <= 0 || head_dim > 64 || head_dim % 8 != 0 || (long long)lk * head_dim > 8192) return UDT_ERR_BAD_SHAPE;
  if (mask && ldmask < lk) return UDT_ERR_BAD_SHAPE;
  MattnParams p;
  p.q = reinterpret_cast<const uint16_t*>(q);
  p.k = reinterpret_cast<const uint16_t*>(k);
  p.v = reinterpret_cast<const uint16_t*>(v);
  p.o = reinterpret_cast<uint16_t*>(o);
  p.mask = mask; p.kpm = kpm;
  p.heads = heads; p.D = head_dim; p.nq = nq; p.lk = lk;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.ldmask = ldmask;
  p.sq = q_bstride; p.sk = k_bstride; p.sv = v_bstride; p.so = o_bstride;
  p.scale = scale;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t smem = ((size_t)2 * lk * (head_dim + 1) + 4 * 64 + 4 * 256) * sizeof(float);
  hipLaunchKernelGGL(mattn_kernel, dim3(heads, batch), dim3(256), smem, s, p);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_softmax_rows(void* x, int64_t rows, int32_t cols, int32_t ld, void* stream) {
  if (!x) return UDT_ERR_BAD_ARG;
  if (rows <= 0 || cols <= 0 || cols % 8 != 0 || ld % 8 != 0 || rows > 0x7fffffffLL) return UDT_ERR_BAD_SHAPE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  UdtProfScope prof(5, s);        // an HBM-bound row pass: not part of the flash-attention class's FLOP accounting
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s,
                     reinterpret_cast<uint16_t*>(x), cols, ld);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}
