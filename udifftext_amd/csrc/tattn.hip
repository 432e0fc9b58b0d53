// tattn.hip — the text cross-attention branch of a transformer block as ONE kernel.
//
// Reference: BasicTransformerBlock.forward  x = t_attn(t_norm(x), context) + x   (sgm/modules/attention.py:314-341) with
// CrossAttention.forward (:140-174): q = to_q(LN(x)); sim = q k^T * d^-1/2 per head; softmax over the L <= 12 context
// tokens; out = to_out(sim v) + bias.
//
// The context (the label embedding) does not change during sampling and has at most 12 tokens, so everything that
// touches it is folded once per batch (udt_tattn_prepare; exact re-association in fp32, SURVEY.md §9b.1 taken further):
//   scores of head h   S_h = LN(x) Wq_h^T K_h^T * scale = LN(x) A_h            A_h  = scale * Wq_h^T K_h^T   [C, 12]
//   output             out = sum_h softmax(S_h) (V_h Wo_h^T) + b               Bm_h = V_h Wo_h^T             [12, C]
// and the LayerNorm is folded as well: with A' = diag(gamma) A,  s_j = sum_k A'_kj,  c_j = sum_k beta_k A_kj,
//   S_j = rstd * (x . A'_j - mean * s_j) + c_j
// so the kernel multiplies the RAW token rows — no normalised copy, no q, no per-head attention launch:
// (round 4: the two tables are stored FRAGMENT-MAJOR — see tattn_prepare_a_kernel — not as the row-major matrices named here)
//   x tile (TT tokens, staged once in LDS: also the residual) -> row mean / rstd -> S = x A' on the MFMAs ->
//   softmax over each head's 16-column group (12 real + 4 padded columns whose c_j = -1e30) -> P (bf16, LDS) ->
//   out^T = Bm^T P^T on the MFMAs -> + bias + x -> store.
// A head occupies 16 columns, so hp = 16 * heads (80 / 160 / 320 for the UNet's 5 / 10 / 20 heads, padded to a multiple
// of 32).  Replaces, per transformer block and sampler step, layernorm + to_q GEMM + xattn + to_out GEMM (+ the
// bias-add launch of the zero-context half: tiles of the first `zero_samples` samples just add the bias).
// FLOPs drop 5x against the unfused chain (K = 12 per head instead of 64); the kernel is bound by reading x once.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>

namespace {

constexpr int TA_THREADS = 512;        // 8 waves: the MFMA phases are chains of L2-latency-bound steps, more waves hide more of it
constexpr int TA_WAVES = TA_THREADS / 64;
// The MFMA phases read their weight-side fragments (A', Bm^T) straight from L2, each once per workgroup: a launch is ONE round of
// workgroups whose length is a chain of L2 round trips (~1.5 us each), so what counts is how many a wave needs.  Round 4: the
// sizes are template parameters (NKT = C / 64 K-tiles, hp = 16 NKS score columns), the A' ring holds up to 10 K-tiles (160
// registers: 2 round trips per score tile at C = 1280 instead of 4), a tile's WHOLE Bm^T row (NKS fragments) is requested before
// its MFMA chain (1 round trip instead of 5), and the first score tile's A' loads are issued before the wave waits for its x rows.

struct TattnParams {
  const uint16_t* x;       // [M, C] bf16
  uint16_t* out;           // [M, C] bf16 (may alias x)
  const uint16_t* A;       // [B][hp][C] bf16: A' rows (K contiguous)
  const float* sc;         // [B][hp][2]: (s_j, c_j)
  const uint16_t* BmT;     // [B][C][hp] bf16
  const float* bias;       // [C] to_out bias
  const uint16_t* zero;
  int M, C, hp, n_tok, zero_samples;
  int nsplit;              // workgroups per token tile: each recomputes the (cheap) scores and owns 1/nsplit of the output channels
  float eps;
  // Q8 instances (udt_tattn_fused_q8, BASELINE config #5): the result again as an MX8 activation (common.h) for the LayerNorm-folded
  // GEGLU projection that consumes it, + the partial row statistics of that LayerNorm, one part per channel split
  uint8_t* q8_out;         // [M, C] e4m3
  uint32_t* q8_scale;      // [C / 128][M]
  float* rowstat_out;      // [nsplit][M][2]
};

UDT_DEVINL f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

// TT tokens per workgroup (64, or 32 for C = 1280 so that the x tile fits in LDS), 8 waves; NKT = C / 64, NKS = hp / 16
template <int TT, int NKT, int NKS, bool Q8 = false>
__global__ void __launch_bounds__(TA_THREADS) tattn_fused_kernel(const TattnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int C = NKT * 64, hp = NKS * 16;
  constexpr int nkt = NKT;                                  // 64-channel K-tiles of the x tile
  constexpr int TA_AHEAD = NKT <= 10 ? NKT : 8;             // A' K-tiles in flight per wave (C = 1280: 8, the register file is the limit)
  char* const xs = smem;                                    // [nkt][TT][128 B], XOR-swizzled 16-byte slots
  float* const stats = reinterpret_cast<float*>(smem + (size_t)nkt * TT * 128);        // [TT][2]: mean, rstd
  const int prs = hp * 2 + 16;                              // padded row stride of the P tile (bytes)
  char* const pl = reinterpret_cast<char*>(stats + TT * 2);                            // [TT][prs]
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int l31 = lane & 31, hi = lane >> 5;
  const long long tok0 = (long long)blockIdx.x * TT;
  const int b = (int)(tok0 / p.n_tok);
  const uint16_t* xg = p.x + tok0 * C;
  uint16_t* og = p.out + tok0 * C;
  float* const rst = reinterpret_cast<float*>(pl + (size_t)TT * prs);                  // Q8: [TT][C / 32][2] per-block row sums
  const int ct_lo = (int)blockIdx.y * (C >> 5) / p.nsplit, ct_hi = ((int)blockIdx.y + 1) * (C >> 5) / p.nsplit;   // this workgroup's 32-channel tiles
  // Q8: per-row sums over this workgroup's channel tiles, in tile order (deterministic) -> part blockIdx.y of rowstat_out
  auto q8_row_stats = [&]() {
    __syncthreads();
    if (tid < TT && tok0 + tid < p.M) {
      float s = 0.f, q = 0.f;
      for (int ct = ct_lo; ct < ct_hi; ++ct) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(rst + ((size_t)tid * (C >> 5) + ct) * 2);
        s += v[0];
        q += v[1];
      }
      f32x2 o = {s, q};
      reinterpret_cast<f32x2*>(p.rowstat_out)[(long long)blockIdx.y * p.M + tok0 + tid] = o;
    }
  };


  // ---- the first score tile's A' fragments: independent of x, requested before anything is waited for -----------------
  const uint16_t* Ab = p.A + (long long)b * hp * C;
  constexpr int jt = hp >> 5;                               // 32-column tiles of the scores
  constexpr int tiles1 = jt * (TT / 32);
  bf16x8_t ar[TA_AHEAD][4];
  const bool live = b >= p.zero_samples;
  if (live && wave < tiles1) {
    const uint16_t* arow = Ab + (long long)(wave / (TT / 32)) * (32 * C) + lane * 8;
#pragma unroll
    for (int u = 0; u < TA_AHEAD; ++u)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) ar[u][ks] = *reinterpret_cast<const bf16x8_t*>(arow + (u * 4 + ks) * 512);
  }

  // ---- x tile -> LDS by LDS-DMA: piece = (K-tile kt, 8-row group rg); lane -> row rg*8 + lane/8, slot lane%8 ----------
  {
    const int pieces = nkt * (TT / 8);
    const int l3 = lane >> 3, ps = lane & 7;
    for (int pc = wave; pc < pieces; pc += TA_WAVES) {
      const int kt = pc / (TT / 8), rg = pc - kt * (TT / 8);
      const int row = rg * 8 + l3;
      const int slot = ps ^ ((row >> 1) & 7);
      const uint16_t* src = (tok0 + row < p.M) ? (xg + (long long)row * C + kt * 64 + slot * 8) : p.zero;
      glds16(src, xs + (size_t)pc * 1024);
    }
    wait_vmcnt0();
  }
  __syncthreads();

  // ---- the zero-context half: x + to_out.bias (reference: k = v = 0 -> attention output 0 -> to_out reduces to its bias)
  if (b < p.zero_samples) {
    const int ch_lo = ct_lo * 4, ch_hi = ct_hi * 4;           // (whole 32-channel tiles: a quad of threads = one MX block)
    const int nch = ch_hi - ch_lo;
    for (int i0 = 0; i0 < TT * nch; i0 += TA_THREADS) {
      const int i = i0 + tid;
      const bool act = i < TT * nch;
      const int row = act ? i / nch : 0, ch = ch_lo + (act ? i - row * nch : 0);
      const bool ok = act && tok0 + row < p.M;
      const int kt = ch >> 3, s = ch & 7;
      const u32x4 v = *reinterpret_cast<const u32x4*>(xs + ((size_t)kt * TT + row) * 128 + ((s ^ ((row >> 1) & 7)) << 4));
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + ch * 8), b1 = *reinterpret_cast<const f32x4*>(p.bias + ch * 8 + 4);
      const float f[8] = {bf16_lo(v[0]) + b0[0], bf16_hi(v[0]) + b0[1], bf16_lo(v[1]) + b0[2], bf16_hi(v[1]) + b0[3],
                          bf16_lo(v[2]) + b1[0], bf16_hi(v[2]) + b1[1], bf16_lo(v[3]) + b1[2], bf16_hi(v[3]) + b1[3]};
      u32x4 o = {pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])};
      if (ok) *reinterpret_cast<u32x4*>(og + (long long)row * C + ch * 8) = o;
      if constexpr (Q8) {
        // (nch % 4 == 0 and TA_THREADS % 4 == 0: the four threads of an aligned quad hold one row's 32-channel block)
        uint32_t sb;
        const u32x2 q8 = mx8_quant_row8(f, sb);
        float ps = 0.f, pq = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) ps += f[j], pq += f[j] * f[j];
        ps = dpp_add<0xB1>(ps); pq = dpp_add<0xB1>(pq);
        ps = dpp_add<0x4E>(ps); pq = dpp_add<0x4E>(pq);
        if (ok) {
          const long long m = tok0 + row;
          *reinterpret_cast<u32x2*>(p.q8_out + m * C + ch * 8) = q8;
          if ((ch & 3) == 0) {
            reinterpret_cast<uint8_t*>(p.q8_scale)[((long long)(ch >> 4) * p.M + m) * 4 + ((ch >> 2) & 3)] = (uint8_t)sb;
            f32x2 st2 = {ps, pq};
            *reinterpret_cast<f32x2*>(rst + ((size_t)row * (C >> 5) + (ch >> 2)) * 2) = st2;
          }
        }
      }
    }
    if constexpr (Q8) q8_row_stats();
    return;
  }

  // ---- LayerNorm statistics of the raw rows: 256 / TT threads per row, fp32 -------------------------------------------
  {
    constexpr int TPR = TA_THREADS / TT;                    // 8 or 16 threads per row (consecutive lanes)
    const int row = tid / TPR, part = tid - row * TPR;
    const int swz = (row >> 1) & 7;
    float s = 0.f, q = 0.f;
    for (int ch = part; ch < nkt * 8; ch += TPR) {
      const int kt = ch >> 3, sl = ch & 7;
      const u32x4 v = *reinterpret_cast<const u32x4*>(xs + ((size_t)kt * TT + row) * 128 + ((sl ^ swz) << 4));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = bf16_lo(v[j]), bb = bf16_hi(v[j]);
        s += a + bb;
        q += a * a + bb * bb;
      }
    }
    s = dpp_add<0xB1>(s); q = dpp_add<0xB1>(q);             // lane ^ 1
    s = dpp_add<0x4E>(s); q = dpp_add<0x4E>(q);             // lane ^ 2
    s += __shfl_xor(s, 4); q += __shfl_xor(q, 4);           // lane ^ 4
    if (TPR == 16) { s += __shfl_xor(s, 8); q += __shfl_xor(q, 8); }
    if (part == 0) {
      const float mean = s / (float)C;
      float var = q / (float)C - mean * mean;
      if (var < 0.f) var = 0.f;
      stats[row * 2] = mean;
      stats[row * 2 + 1] = rsqrtf(var + p.eps);
    }
  }
  __syncthreads();

  // ---- S^T = A' x^T  (D rows = score columns j, D cols = tokens: a lane owns one token) -> softmax -> P in LDS ---------
  const float* scb = p.sc + (long long)b * hp * 2;
  for (int t = wave; t < tiles1; t += TA_WAVES) {
    const int rt = t % (TT / 32), ct = t / (TT / 32);
    const int row = rt * 32 + l31;                          // this lane's token (as MFMA column)
    const int swz = (l31 >> 1) & 7;                         // (row >> 1) & 7 with row = rt*32 + l31
    const uint16_t* arow = Ab + (long long)ct * (32 * C) + lane * 8;          // fragment-major tables: 1 KiB per (K-tile, k-step)
    f32x16 acc = zero16();
    // the A' fragments come straight from L2 (each is used once per workgroup): loads run TA_AHEAD K-tiles ahead of the MFMAs
    // (a register ring, the K loop fully unrolled; the first tile's ring was filled at kernel entry)
    if (t != wave) {
#pragma unroll
      for (int u = 0; u < TA_AHEAD; ++u)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ar[u][ks] = *reinterpret_cast<const bf16x8_t*>(arow + (u * 4 + ks) * 512);
    }
#pragma unroll
    for (int kt = 0; kt < nkt; ++kt) {
      const int u = kt % TA_AHEAD;
      const char* xrow = xs + ((size_t)kt * TT + row) * 128;
      bf16x8_t af[4], xf[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        af[ks] = ar[u][ks];
        xf[ks] = lds_read_frag(xrow + (((ks * 2 + hi) ^ swz) << 4));
      }
      if (kt + TA_AHEAD < nkt) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ar[u][ks] = *reinterpret_cast<const bf16x8_t*>(arow + ((kt + TA_AHEAD) * 4 + ks) * 512);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc = mfma32(af[ks], xf[ks], acc);
    }
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    // reg r <-> column j = ct*32 + (r&3) + 8*(r>>2) + 4*hi; regs 0..7 lie in the tile's first head, 8..15 in the second
    float sv[16];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int j = ct * 32 + 8 * r4 + 4 * hi;
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(scb + j * 2), c1 = *reinterpret_cast<const f32x4*>(scb + j * 2 + 4);
      sv[r4 * 4 + 0] = rstd * (acc[r4 * 4 + 0] - mean * c0[0]) + c0[1];
      sv[r4 * 4 + 1] = rstd * (acc[r4 * 4 + 1] - mean * c0[2]) + c0[3];
      sv[r4 * 4 + 2] = rstd * (acc[r4 * 4 + 2] - mean * c1[0]) + c1[1];
      sv[r4 * 4 + 3] = rstd * (acc[r4 * 4 + 3] - mean * c1[2]) + c1[3];
    }
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {                        // the two heads of this 32-column tile
      float mx = sv[h2 * 8];
#pragma unroll
      for (int r = 1; r < 8; ++r) mx = fmaxf(mx, sv[h2 * 8 + r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        sv[h2 * 8 + r] = __expf(sv[h2 * 8 + r] - mx);
        sum += sv[h2 * 8 + r];
      }
      sum += __shfl_xor(sum, 32);
      const float inv = 1.0f / sum;
#pragma unroll
      for (int r = 0; r < 8; ++r) sv[h2 * 8 + r] *= inv;
    }
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int j = ct * 32 + 8 * r4 + 4 * hi;
      u32x2 pk = {pack_bf16x2(sv[r4 * 4], sv[r4 * 4 + 1]), pack_bf16x2(sv[r4 * 4 + 2], sv[r4 * 4 + 3])};
      *reinterpret_cast<u32x2*>(pl + (size_t)row * prs + j * 2) = pk;
    }
  }
  __syncthreads();

  // ---- out^T = Bm^T P^T  (D rows = channels, D cols = tokens) + bias + residual -> store ---------------------------------
  const uint16_t* Bb = p.BmT + (long long)b * C * hp;
  const int tiles2 = (ct_hi - ct_lo) * (TT / 32);
  constexpr int nks = NKS;                                  // 16-wide k-steps over the score columns
  for (int t = wave; t < tiles2; t += TA_WAVES) {
    const int rt = t % (TT / 32), ct = ct_lo + t / (TT / 32);
    const int row = rt * 32 + l31;
    const uint16_t* brow = Bb + (long long)ct * (32 * hp) + lane * 8;
    const char* prow = pl + (size_t)row * prs + hi * 16;
    f32x16 acc = zero16();
    bf16x8_t bn[NKS];                                       // the tile's whole Bm^T row: one L2 round trip in front of the chain
#pragma unroll
    for (int i = 0; i < NKS; ++i) bn[i] = *reinterpret_cast<const bf16x8_t*>(brow + i * 512);
#pragma unroll
    for (int ks = 0; ks < nks; ++ks) {
      const bf16x8_t pf = *reinterpret_cast<const bf16x8_t*>(prow + ks * 32);
      acc = mfma32(bn[ks], pf, acc);
    }
    {
      const bool rok = tok0 + row < p.M;                    // (rows past M stage the zero page: every lane runs the cross-lane steps)
      const int swz = (row >> 1) & 7;
      float v[16];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int c = ct * 32 + 8 * r4 + 4 * hi;            // 4 consecutive channels
        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + c);
        const int kt = c >> 6, sl = (c & 63) >> 3;
        const u32x2 xr = *reinterpret_cast<const u32x2*>(xs + ((size_t)kt * TT + row) * 128 + ((sl ^ swz) << 4) + (c & 7) * 2);
        v[r4 * 4 + 0] = acc[r4 * 4 + 0] + bv[0] + bf16_lo(xr[0]);
        v[r4 * 4 + 1] = acc[r4 * 4 + 1] + bv[1] + bf16_hi(xr[0]);
        v[r4 * 4 + 2] = acc[r4 * 4 + 2] + bv[2] + bf16_lo(xr[1]);
        v[r4 * 4 + 3] = acc[r4 * 4 + 3] + bv[3] + bf16_hi(xr[1]);
        u32x2 pk = {pack_bf16x2(v[r4 * 4 + 0], v[r4 * 4 + 1]), pack_bf16x2(v[r4 * 4 + 2], v[r4 * 4 + 3])};
        if (rok) *reinterpret_cast<u32x2*>(og + (long long)row * C + c) = pk;
      }
      if constexpr (Q8) {
        // this 32-channel tile of the row is one MX block: 16 of its values here, 16 in lane ^ 32
        uint32_t q[4], sb;
        mx8_quant_acc16(v, q, sb);
        float ps = 0.f, pq = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) ps += v[r], pq += v[r] * v[r];
        ps = xor32_sum(ps);
        pq = xor32_sum(pq);
        if (rok) {
          const long long m = tok0 + row;
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) *reinterpret_cast<uint32_t*>(p.q8_out + m * C + ct * 32 + 8 * r4 + 4 * hi) = q[r4];
          if (hi == 0) {
            reinterpret_cast<uint8_t*>(p.q8_scale)[((long long)(ct >> 2) * p.M + m) * 4 + (ct & 3)] = (uint8_t)sb;
            f32x2 st2 = {ps, pq};
            *reinterpret_cast<f32x2*>(rst + ((size_t)row * (C >> 5) + ct) * 2) = st2;
          }
        }
      }
    }
  }
  if constexpr (Q8) q8_row_stats();
}

// ---- once per batch: fold the context into the per-sample tables ------------------------------------------------------
// grid (hp, B): one workgroup per score column j = 16 h + i.  A'[b][j][k] = gamma_k * scale * sum_d Wq[h*64+d][k] K[b][i][h*64+d];
// s_j = sum_k bf16(A'), c_j = scale * sum_k beta_k sum_d (...); padded columns (i >= L): A' = 0, s = 0, c = -1e30.
__global__ void __launch_bounds__(256) tattn_prepare_a_kernel(const uint16_t* __restrict__ kv, int ldkv, const uint16_t* __restrict__ wq,
                                                              int ldwq, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              uint16_t* __restrict__ A, float* __restrict__ sc, int L, int C, int hp,
                                                              int heads, float scale) {
  __shared__ float kvec[64];
  __shared__ float red[2][4];
  const int j = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const int h = j >> 4, i = j & 15;
  // FRAGMENT-MAJOR layout (round 4): the 16 bytes a lane of tattn_fused_kernel loads for (score tile j / 32, K-tile k / 64,
  // k-step) are contiguous with its neighbours' — every fragment load is one 1-KiB run of whole cache lines
  uint16_t* const Ab = A + (long long)b * hp * C;
  const int nkt = C >> 6;
  auto aoff = [&](int k) {
    return ((((long long)(j >> 5) * nkt + (k >> 6)) * 4 + ((k >> 4) & 3)) * 64 + ((k >> 3) & 1) * 32 + (j & 31)) * 8 + (k & 7);
  };
  float* scj = sc + ((long long)b * hp + j) * 2;
  if (h >= heads || i >= L) {
    for (int k = t; k < C; k += 256) Ab[aoff(k)] = 0;
    if (t == 0) { scj[0] = 0.f; scj[1] = -1e30f; }
    return;
  }
  if (t < 64) kvec[t] = bf16_bits_to_f32(kv[((long long)b * L + i) * ldkv + h * 64 + t]) * scale;
  __syncthreads();
  float s = 0.f, c = 0.f;
  for (int k = t; k < C; k += 256) {
    float a = 0.f;
#pragma unroll 8
    for (int d = 0; d < 64; ++d) a += bf16_bits_to_f32(wq[(long long)(h * 64 + d) * ldwq + k]) * kvec[d];
    const uint32_t pk = pack_bf16x2(a * gamma[k], 0.f);
    Ab[aoff(k)] = (uint16_t)(pk & 0xffffu);
    s += bf16_lo(pk);
    c += beta[k] * a;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); c += __shfl_xor(c, off); }
  if ((t & 63) == 0) { red[0][t >> 6] = s; red[1][t >> 6] = c; }
  __syncthreads();
  if (t == 0) {
    scj[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    scj[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

// BmT[b][c][j] = sum_d V[b][i][h*64+d] Wo[c][h*64+d]   (j = 16 h + i; padded columns 0).  One thread per element.
__global__ void __launch_bounds__(256) tattn_prepare_b_kernel(const uint16_t* __restrict__ kv, int ldkv, int v_off,
                                                              const uint16_t* __restrict__ wo, int ldwo, uint16_t* __restrict__ BmT,
                                                              int B, int L, int C, int hp, int heads) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)B * C * hp) return;
  const int j = (int)(idx % hp);
  const long long bc = idx / hp;
  const int c = (int)(bc % C), b = (int)(bc / C);
  const int h = j >> 4, i = j & 15;
  float a = 0.f;
  if (h < heads && i < L) {
    const uint16_t* vr = kv + ((long long)b * L + i) * ldkv + v_off + h * 64;
    const uint16_t* wr = wo + (long long)c * ldwo + h * 64;
#pragma unroll 8
    for (int d = 0; d < 64; ++d) a += bf16_bits_to_f32(vr[d]) * bf16_bits_to_f32(wr[d]);
  }
  // fragment-major: (channel tile c / 32, k-step j / 16) -> 64 lanes x 16 bytes
  const long long off = ((((long long)(c >> 5) * (hp >> 4) + (j >> 4)) * 64) + ((j >> 3) & 1) * 32 + (c & 31)) * 8 + (j & 7);
  BmT[(long long)b * C * hp + off] = (uint16_t)(pack_bf16x2(a, 0.f) & 0xffffu);
}

}  // namespace

extern "C" int32_t udt_tattn_hp(int32_t heads) { return ((heads * 16 + 31) / 32) * 32; }

extern "C" int udt_tattn_prepare(const void* kv, int32_t ldkv, const void* wq, int32_t ldwq, const void* wo, int32_t ldwo,
                                 const float* gamma, const float* beta, void* A, float* sc, void* BmT, int32_t B, int32_t L,
                                 int32_t C, int32_t heads, float scale, void* stream) {
  if (!kv || !wq || !wo || !gamma || !beta || !A || !sc || !BmT) return UDT_ERR_BAD_ARG;
  if (B <= 0 || L <= 0 || L > 12 || heads <= 0 || C != heads * 64 || ldkv < 2 * C || ldwq < C || ldwo < C) return UDT_ERR_BAD_SHAPE;
  const int hp = udt_tattn_hp(heads);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  UdtProfScope prof(3, s);
  hipLaunchKernelGGL(tattn_prepare_a_kernel, dim3(hp, B), dim3(256), 0, s, reinterpret_cast<const uint16_t*>(kv), ldkv,
                     reinterpret_cast<const uint16_t*>(wq), ldwq, gamma, beta, reinterpret_cast<uint16_t*>(A), sc, L, C, hp, heads, scale);
  UDT_CHECK_LAUNCH();
  const long long n = (long long)B * C * hp;
  hipLaunchKernelGGL(tattn_prepare_b_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                     reinterpret_cast<const uint16_t*>(kv), ldkv, C, reinterpret_cast<const uint16_t*>(wo), ldwo,
                     reinterpret_cast<uint16_t*>(BmT), B, L, C, hp, heads);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

namespace {
// workgroups per token tile (channel splits): few token tiles (the 16x16 / 8x8 levels) split the output channels over up to 8
// workgroups per tile so that the launch covers the chip; every split recomputes the statistics and scores of its tile
// (round 3: a full chip of workgroups — the launch is a chain of L2-latency-bound steps, so idle CUs are the one thing that is free)
// tokens per workgroup: 32 for C = 1280 (the x tile must fit in LDS) and, since round 5, for C = 640 (two workgroups per CU: 17.6 ->
// 15.1 us on 8 x 1024 tokens; C = 320 measured the same either way, 15.5 / 15.8 us, and keeps 64)
int tattn_tt(int C) { return C >= 640 ? 32 : 64; }
int tattn_nsplit(int M, int TT) {
  const unsigned grid = (unsigned)(M / TT);
  constexpr unsigned target = 256u;
  int nsplit = 1;
  while (nsplit < 8 && grid * nsplit < target) nsplit *= 2;
  return nsplit;
}

int tattn_launch(const void* x, void* out, const void* A, const float* sc, const void* BmT, const float* bias, int32_t B, int32_t n_tok,
                 int32_t C, int32_t heads, int32_t zero_samples, float eps, void* q8_out, void* q8_scale, float* rowstat_out, void* stream) {
  if (!x || !out || !bias || (zero_samples < B && (!A || !sc || !BmT))) return UDT_ERR_BAD_ARG;
  if (B <= 0 || n_tok <= 0 || heads <= 0 || C != heads * 64 || C > 1280 || zero_samples < 0 || zero_samples > B) return UDT_ERR_BAD_SHAPE;
  const int TT = tattn_tt(C);
  if (n_tok % TT != 0) return UDT_ERR_BAD_SHAPE;             // a token tile lies in one sample
  const bool q8 = q8_out != nullptr;
  if (q8 && (!q8_scale || !rowstat_out || C % 128 != 0)) return UDT_ERR_BAD_ARG;
  TattnParams p;
  p.x = reinterpret_cast<const uint16_t*>(x); p.out = reinterpret_cast<uint16_t*>(out);
  p.A = reinterpret_cast<const uint16_t*>(A); p.sc = sc; p.BmT = reinterpret_cast<const uint16_t*>(BmT); p.bias = bias;
  p.zero = udt_zero_page();
  if (!p.zero) return UDT_ERR_HIP;
  p.M = B * n_tok; p.C = C; p.hp = udt_tattn_hp(heads); p.n_tok = n_tok; p.zero_samples = zero_samples; p.eps = eps;
  p.q8_out = reinterpret_cast<uint8_t*>(q8_out); p.q8_scale = reinterpret_cast<uint32_t*>(q8_scale); p.rowstat_out = rowstat_out;
  const size_t smem = (size_t)TT * C * 2 + (size_t)TT * 2 * sizeof(float) + (size_t)TT * (p.hp * 2 + 16) + (q8 ? (size_t)TT * (C / 32) * 8 : 0);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  UdtProfScope prof(3, s);
  if (prof.rec) {
    char tag[96];
    snprintf(tag, sizeof(tag), "tattn_fused%s B=%d n=%d C=%d zero=%d", q8 ? "+q8" : "", B, n_tok, C, zero_samples);
    udt_prof_tag(prof.rec, tag);
  }
  static bool attr_done[5] = {false, false, false, false, false};   // (max dynamic LDS; set once per process — one device per process)
  const unsigned grid = (unsigned)(p.M / TT);
  p.nsplit = tattn_nsplit(p.M, TT);
  // instances: the UNet's three widths (C = 320 / 640 / 1280 with 5 / 10 / 20 heads: hp = 80 -> padded 96, 160, 320); the MX8-emitting
  // form for the two widths whose linears run on e4m3 operands in config #5
  const void* fn = nullptr;
  int which = -1;
  if (C == 320 && p.hp == 96 && !q8) { fn = reinterpret_cast<const void*>(tattn_fused_kernel<64, 5, 6>); which = 0; }
  else if (C == 640 && p.hp == 160) { fn = q8 ? reinterpret_cast<const void*>(tattn_fused_kernel<32, 10, 10, true>) : reinterpret_cast<const void*>(tattn_fused_kernel<32, 10, 10>); which = q8 ? 3 : 1; }
  else if (C == 1280 && p.hp == 320) { fn = q8 ? reinterpret_cast<const void*>(tattn_fused_kernel<32, 20, 20, true>) : reinterpret_cast<const void*>(tattn_fused_kernel<32, 20, 20>); which = q8 ? 4 : 2; }
  else return UDT_ERR_BAD_SHAPE;
  if (!attr_done[which]) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (e != hipSuccess) return udt_set_hip_error(e);
    attr_done[which] = true;
  }
  void* args[] = {&p};
  hipError_t el = hipLaunchKernel(fn, dim3(grid, p.nsplit), dim3(TA_THREADS), args, smem, s);
  if (el != hipSuccess) return udt_set_hip_error(el);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}
}  // namespace

extern "C" int udt_tattn_fused(const void* x, void* out, const void* A, const float* sc, const void* BmT, const float* bias,
                               int32_t B, int32_t n_tok, int32_t C, int32_t heads, int32_t zero_samples, float eps, void* stream) {
  return tattn_launch(x, out, A, sc, BmT, bias, B, n_tok, C, heads, zero_samples, eps, nullptr, nullptr, nullptr, stream);
}

extern "C" int32_t udt_tattn_rowstat_parts(int32_t B, int32_t n_tok, int32_t C) {
  if (B <= 0 || n_tok <= 0 || (C != 640 && C != 1280)) return 0;
  const int TT = tattn_tt(C);
  if (n_tok % TT != 0) return 0;
  return tattn_nsplit(B * n_tok, TT);
}

extern "C" int udt_tattn_fused_q8(const void* x, void* out, const void* A, const float* sc, const void* BmT, const float* bias,
                                  int32_t B, int32_t n_tok, int32_t C, int32_t heads, int32_t zero_samples, float eps, void* q8_out,
                                  void* q8_scale, float* rowstat_out, void* stream) {
  if (!q8_out) return UDT_ERR_BAD_ARG;
  return tattn_launch(x, out, A, sc, BmT, bias, B, n_tok, C, heads, zero_samples, eps, q8_out, q8_scale, rowstat_out, stream);
}
