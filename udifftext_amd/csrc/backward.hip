// backward.hip — dX-only backward kernels for attend-and-excite (first slice: one ResBlock + one SpatialTransformer).
//
// Reference: EulerEDMSampler.attend_and_excite (sgm/modules/diffusionmodules/sampling.py:233-252) takes
// torch.autograd.grad(local_loss, x) through the UNet; the loss is FullLoss.get_min_local_loss (loss.py:192-235) on the t_attn
// probability maps.  Only the gradient with respect to ACTIVATIONS is needed (x is updated, the weights are frozen), so:
//   * linears and convolutions run their backward-data as FORWARD launches of the existing GEMM / convolution kernels on re-packed
//     weights (W^T; 180-degree rotated taps with the channel roles swapped) — nothing of them lives here;
//   * this file holds what has no forward twin:
//       udt_attn_bwd          flash-attention backward (head_dim 64): dQ, dK, dV from Q, K, V, O, dO — two launches of one MFMA kernel
//                             template (the "owner" tile keeps its fragments in registers, the other side streams through LDS; no
//                             atomics: dQ is owned by query tiles, dK / dV by key tiles; S is recomputed, never stored)
//       udt_xattn_bwd         text cross-attention (<= 16 context tokens): dq from the probability gradient (local loss) and dO
//       udt_local_loss_bwd    d(-min_l max_n(mask * blur(mean_h P))) / dP: a 3x3 stencil around the arg-max of the arg-min token
//       udt_layernorm_bwd     LayerNorm backward (dX), optional fused add of the gradient that arrives over the residual path
//       udt_gn_bwd            GroupNorm(32) (+ SiLU) backward (dX), statistics recomputed from x, optional fused add
//       udt_geglu_fwd / _bwd  GEGLU on stored pre-activations (the fused GEMM epilogue of the inference path keeps none)
// bf16 storage, fp32 arithmetic; every kernel is deterministic (fixed reduction orders, no atomics).
#include "common.h"
#include <math.h>
#include <stdlib.h>

namespace {

// ------------------------------------------------------------------------------------------------ flash attention backward
// mfma32(A, B, acc): lane (x, h) of A holds A[i = x][kk = 8 h .. 8 h + 7], of B holds B[j = x][kk = 8 h ..]; the lane (l31, hi) gets
// acc[r] = sum_kk A[i = 8 (r >> 2) + 4 hi + (r & 3)][kk] * B[j = l31][kk]   (the layout attention.hip's forward kernels rely on).
// One template serves both launches:
//   DKV = false ("dq"): owner = 128 queries (Q, dO fragments in registers, LSE and D = rowsum(dO o O) per lane), streamed = keys
//                       (K, V tiles of 32 rows through LDS).  Pass 1 computes LSE (log2 domain) and D and stores them for the other
//                       launch; pass 2 accumulates dQ^T[d][q] += sum_k K^T[d][k] dS[k][q].
//   DKV = true  ("dkv"): owner = 128 keys (K, V fragments in registers), streamed = queries (Q, dO tiles + their LSE, D):
//                       dV^T[d][k] += sum_q dO^T[d][q] P[q][k],  dK^T[d][k] += sum_q Q^T[d][q] dS[q][k].
// In both, the lane dimension of the 32 x 32 score tile is the OWNER index and the register dimension the STREAMED index, so P / dS
// leave the first MFMAs in exactly the B-operand order of the second ones (contraction over the streamed index); the matching A
// operands are the streamed tiles transposed, which the staging code writes next to the row-major copy.
struct AttnBwdParams {
  const uint16_t* q;
  const uint16_t* k;
  const uint16_t* v;
  const uint16_t* o;
  const uint16_t* d_o;
  uint16_t* dq;
  uint16_t* dk;
  uint16_t* dv;
  float* lse;              // [batch * heads][n], log2 domain: m + log2(sum)
  float* dsum;             // [batch * heads][n]
  int heads, n, tiles;
  int ldq, ldo, ldd;       // row strides (elements) of q|k|v, of o / d_o, of dq|dk|dv
  long long sq, so, sd;    // batch strides
  float scale, scale_log2e;
};

constexpr int AB_XP = 72;  // row-major pitch (elements): 144 B, rows shift by 4 banks
constexpr int AB_TP = 40;  // transposed pitch (elements): 80 B

UDT_DEVINL bf16x8_t ab_load8(const uint16_t* p, bool ok) {
  u32x4 z = {0u, 0u, 0u, 0u};
  if (ok) z = *reinterpret_cast<const u32x4*>(p);
  return __builtin_bit_cast(bf16x8_t, z);
}

// AB_PF: streamed tiles whose global loads are in flight (a register ring)
// FULL: n is a multiple of 128 (every UNet level): no bounds tests on the scores
template <bool DKV, int AB_PF, bool FULL>
__global__ void __launch_bounds__(256) attn_bwd_kernel(const AttnBwdParams p) {
  __shared__ __attribute__((aligned(16))) uint16_t x1[32 * AB_XP];
  __shared__ __attribute__((aligned(16))) uint16_t x2[32 * AB_XP];
  __shared__ __attribute__((aligned(16))) uint16_t x1t[64 * AB_TP];
  __shared__ __attribute__((aligned(16))) uint16_t x2t[64 * AB_TP];
  __shared__ __attribute__((aligned(16))) float sc_lse[32];
  __shared__ __attribute__((aligned(16))) float sc_d[32];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int bh = blockIdx.x / p.tiles;
  const int tile = blockIdx.x - bh * p.tiles;
  const int b = bh / p.heads, h = bh - b * p.heads;
  const uint16_t* Q = p.q + (long long)b * p.sq + h * 64;
  const uint16_t* K = p.k + (long long)b * p.sq + h * 64;
  const uint16_t* V = p.v + (long long)b * p.sq + h * 64;
  const uint16_t* O = p.o + (long long)b * p.so + h * 64;
  const uint16_t* DO = p.d_o + (long long)b * p.so + h * 64;
  float* LSE = p.lse + (long long)bh * p.n;
  float* DS = p.dsum + (long long)bh * p.n;
  const int orow = tile * 128 + wave * 32 + l31;
  const bool ook = orow < p.n;
  const float c = p.scale_log2e;

  // owner fragments (B operands): 8 consecutive head dims at ks * 16 + hi * 8 of this lane's row
  const uint16_t* Y1 = DKV ? K : Q;
  const uint16_t* Y2 = DKV ? V : DO;
  const int ldy2 = DKV ? p.ldq : p.ldo;
  bf16x8_t y1f[4], y2f[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    y1f[ks] = ab_load8(Y1 + (long long)orow * p.ldq + ks * 16 + hi * 8, ook);
    y2f[ks] = ab_load8(Y2 + (long long)orow * ldy2 + ks * 16 + hi * 8, ook);
  }
  const uint16_t* X1 = DKV ? Q : K;
  const uint16_t* X2 = DKV ? DO : V;
  const int ldx2 = DKV ? p.ldo : p.ldq;
  const int srow = tid >> 3, sch = tid & 7;
  const int nst = (p.n + 31) / 32;

  // one streamed tile -> LDS: row-major (A fragments of S / dP) and transposed (A fragments of the output products).  The global
  // loads run AB_PF tiles ahead in a register ring (8 VGPRs per slot): a batch-1 launch has at most one workgroup per CU, so nothing
  // else hides the L2 / HBM latency of a tile (load -> LDS -> barrier in line: 138 us per 4096-token launch, 2.4x the ring's)
  u32x4 ra[AB_PF], rb[AB_PF];
  float rl[AB_PF], rd[AB_PF];
  auto gload = [&](int st, int slot, bool second) {
    const int s = st * 32 + srow;
    const bool ok = s < p.n && st < nst;
    u32x4 a = {0u, 0u, 0u, 0u}, bq = {0u, 0u, 0u, 0u};
    if (ok) a = *reinterpret_cast<const u32x4*>(X1 + (long long)s * p.ldq + sch * 8);
    if (ok && second) bq = *reinterpret_cast<const u32x4*>(X2 + (long long)s * ldx2 + sch * 8);
    ra[slot] = a;
    rb[slot] = bq;
    if (DKV && second) {
      const int s2 = st * 32 + (tid & 31);
      const bool ok2 = s2 < p.n && st < nst;
      rl[slot] = ok2 ? LSE[s2] : 0.f;
      rd[slot] = ok2 ? DS[s2] : 0.f;
    }
  };
  auto lstore = [&](int slot, bool second, bool transposed) {
    const u32x4 a = ra[slot], bq = rb[slot];
    *reinterpret_cast<u32x4*>(x1 + srow * AB_XP + sch * 8) = a;
    if (second) *reinterpret_cast<u32x4*>(x2 + srow * AB_XP + sch * 8) = bq;
    if (transposed) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x1t[(sch * 8 + 2 * j) * AB_TP + srow] = (uint16_t)(a[j] & 0xffffu);
        x1t[(sch * 8 + 2 * j + 1) * AB_TP + srow] = (uint16_t)(a[j] >> 16);
        if (DKV) {
          x2t[(sch * 8 + 2 * j) * AB_TP + srow] = (uint16_t)(bq[j] & 0xffffu);
          x2t[(sch * 8 + 2 * j + 1) * AB_TP + srow] = (uint16_t)(bq[j] >> 16);
        }
      }
    }
    if (DKV && second && tid < 32) {
      sc_lse[tid] = rl[slot];
      sc_d[tid] = rd[slot];
    }
  };
  auto scores = [&](const uint16_t* xs, const bf16x8_t (&yf)[4]) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(xs + l31 * AB_XP + ks * 16 + hi * 8);
      acc = mfma32(a, yf[ks], acc);
    }
    return acc;
  };
  // A fragment of a transposed tile: rows d = it * 32 + l31, the 8 streamed indices of k-step `half` in the order P / dS sit in
  auto tfrag = [&](const uint16_t* xt, int it, int half) {
    const uint16_t* base = xt + (it * 32 + l31) * AB_TP + 16 * half + 4 * hi;
    const u32x2 lo = *reinterpret_cast<const u32x2*>(base);
    const u32x2 hh = *reinterpret_cast<const u32x2*>(base + 8);
    const u32x4 v4 = {lo[0], lo[1], hh[0], hh[1]};
    return __builtin_bit_cast(bf16x8_t, v4);
  };
  auto pack8 = [](const float* f) {
    const u32x4 v4 = {pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])};
    return __builtin_bit_cast(bf16x8_t, v4);
  };

  float lse_o = 0.f, d_o_sum = 0.f;
  if constexpr (!DKV) {
    // ---- pass 1: LSE of this lane's query over all keys (the two half-waves see different keys: merged at the end), and D
    float m_run = -INFINITY, l_run = 0.f;
#pragma unroll
    for (int u = 0; u < AB_PF; ++u) gload(u, u, false);
    for (int st0 = 0; st0 < nst; st0 += AB_PF)
#pragma unroll
    for (int u = 0; u < AB_PF; ++u) {
      const int st = st0 + u;
      if (st >= nst) break;
      __syncthreads();
      lstore(u, false, false);
      __syncthreads();
      gload(st + AB_PF, u, false);
      const f32x16 s = scores(x1, y1f);
      float sv[16];
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = st * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
        sv[r] = (FULL || key < p.n) ? s[r] * c : -INFINITY;
        mx = fmaxf(mx, sv[r]);
      }
      if (mx > -INFINITY) {
        const float m_new = fmaxf(m_run, mx);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += fast_exp2(sv[r] - m_new);
        l_run = l_run * fast_exp2(m_run - m_new) + sum;
        m_run = m_new;
      }
    }
    const float m_oth = __shfl_xor(m_run, 32), l_oth = __shfl_xor(l_run, 32);
    const float m_all = fmaxf(m_run, m_oth);
    float l_all = 0.f;
    if (m_run > -INFINITY) l_all += l_run * fast_exp2(m_run - m_all);
    if (m_oth > -INFINITY) l_all += l_oth * fast_exp2(m_oth - m_all);
    lse_o = m_all + __log2f(l_all);
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const u32x4 ov = __builtin_bit_cast(u32x4, ab_load8(O + (long long)orow * p.ldo + ks * 16 + hi * 8, ook));
      const u32x4 gv = __builtin_bit_cast(u32x4, y2f[ks]);
#pragma unroll
      for (int j = 0; j < 4; ++j) part += bf16_lo(ov[j]) * bf16_lo(gv[j]) + bf16_hi(ov[j]) * bf16_hi(gv[j]);
    }
    d_o_sum = part + __shfl_xor(part, 32);
    if (ook && hi == 0) {
      LSE[orow] = lse_o;
      DS[orow] = d_o_sum;
    }
  }

  f32x16 acc1[2], acc2[2];             // dq: acc1 = dQ^T; dkv: acc1 = dK^T, acc2 = dV^T   ([d tile][...])
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[it][r] = acc2[it][r] = 0.f;

#pragma unroll
  for (int u = 0; u < AB_PF; ++u) gload(u, u, true);
  for (int st0 = 0; st0 < nst; st0 += AB_PF)
#pragma unroll
  for (int u = 0; u < AB_PF; ++u) {
    const int st = st0 + u;
    if (st >= nst) break;
    __syncthreads();
    lstore(u, true, true);
    __syncthreads();
    gload(st + AB_PF, u, true);
    const f32x16 s = scores(x1, y1f);
    const f32x16 dp = scores(x2, y2f);
    float pv[16], dsv[16];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      f32x4 ls = {lse_o, lse_o, lse_o, lse_o}, dd = {d_o_sum, d_o_sum, d_o_sum, d_o_sum};
      if constexpr (DKV) {
        ls = *reinterpret_cast<const f32x4*>(sc_lse + 8 * q4 + 4 * hi);
        dd = *reinterpret_cast<const f32x4*>(sc_d + 8 * q4 + 4 * hi);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = q4 * 4 + e;
        const int sidx = st * 32 + 8 * q4 + 4 * hi + e;
        const float pr = (FULL || (sidx < p.n && ook)) ? fast_exp2(s[r] * c - ls[e]) : 0.f;
        pv[r] = pr;
        dsv[r] = pr * (dp[r] - dd[e]);                          // (the softmax scale multiplies the finished dQ / dK sums)
      }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const bf16x8_t dsf = pack8(dsv + half * 8);
#pragma unroll
      for (int it = 0; it < 2; ++it) acc1[it] = mfma32(tfrag(x1t, it, half), dsf, acc1[it]);
      if constexpr (DKV) {
        const bf16x8_t pf = pack8(pv + half * 8);
#pragma unroll
        for (int it = 0; it < 2; ++it) acc2[it] = mfma32(tfrag(x2t, it, half), pf, acc2[it]);
      }
    }
  }
  if (ook) {
    uint16_t* o1 = (DKV ? p.dk : p.dq) + (long long)b * p.sd + (long long)orow * p.ldd + h * 64;
    uint16_t* o2 = p.dv + (long long)b * p.sd + (long long)orow * p.ldd + h * 64;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int d = it * 32 + 8 * q4 + 4 * hi;
        const u32x2 a = {pack_bf16x2(acc1[it][q4 * 4] * p.scale, acc1[it][q4 * 4 + 1] * p.scale),
                         pack_bf16x2(acc1[it][q4 * 4 + 2] * p.scale, acc1[it][q4 * 4 + 3] * p.scale)};
        *reinterpret_cast<u32x2*>(o1 + d) = a;
        if constexpr (DKV) {
          const u32x2 bb = {pack_bf16x2(acc2[it][q4 * 4], acc2[it][q4 * 4 + 1]), pack_bf16x2(acc2[it][q4 * 4 + 2], acc2[it][q4 * 4 + 3])};
          *reinterpret_cast<u32x2*>(o2 + d) = bb;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------ text cross-attention backward
// P = softmax_l(scale q K^T) (or the sigmoid for a single context token, reference attention.py:159-162), out = P V.  Given the
// gradient of the loss with respect to P (the local loss reads the probabilities directly) and / or with respect to out:
//   g_l = dP_l + sum_d dO_d V[l][d];   dS_l = P_l (g_l - sum_j P_j g_j)   (sigmoid: P (1 - P) g);   dq_d = scale sum_l dS_l K[l][d].
// The context is a constant of the problem (the label embedding): no dK / dV.  One lane per (batch, head, query), K and V of the
// head as fp32 in LDS — the forward kernel's arrangement.
struct XattnBwdParams {
  const uint16_t* k;
  const uint16_t* v;
  const float* probs;
  const float* d_probs;
  const uint16_t* d_o;
  uint16_t* dq;
  int heads, nq, L, ldkv, ldo, lddq;
  float scale;
};
constexpr int XB_L = 16;

__global__ void __launch_bounds__(256) xattn_bwd_kernel(const XattnBwdParams p) {
  __shared__ float ks[XB_L * 64];
  __shared__ float vs[XB_L * 64];
  const int h = blockIdx.y, b = blockIdx.z;
  for (int i = threadIdx.x; i < p.L * 64; i += 256) {
    const int l = i >> 6, d = i & 63;
    const long long off = ((long long)b * p.L + l) * p.ldkv + h * 64 + d;
    ks[i] = bf16_bits_to_f32(p.k[off]);
    vs[i] = bf16_bits_to_f32(p.v[off]);
  }
  __syncthreads();
  const int qi = blockIdx.x * 256 + threadIdx.x;
  if (qi >= p.nq) return;
  const long long prow = (((long long)b * p.heads + h) * p.nq + qi) * p.L;
  float pr[XB_L], g[XB_L];
#pragma unroll
  for (int l = 0; l < XB_L; ++l) {
    pr[l] = l < p.L ? p.probs[prow + l] : 0.f;
    g[l] = (l < p.L && p.d_probs) ? p.d_probs[prow + l] : 0.f;
  }
  if (p.d_o) {
    const uint16_t* gr = p.d_o + ((long long)b * p.nq + qi) * p.ldo + h * 64;
    for (int d0 = 0; d0 < 64; d0 += 8) {
      const u32x4 gv = *reinterpret_cast<const u32x4*>(gr + d0);
      const float gf[8] = {bf16_lo(gv[0]), bf16_hi(gv[0]), bf16_lo(gv[1]), bf16_hi(gv[1]), bf16_lo(gv[2]), bf16_hi(gv[2]), bf16_lo(gv[3]), bf16_hi(gv[3])};
#pragma unroll
      for (int l = 0; l < XB_L; ++l)
        if (l < p.L) {
          const float* vr = vs + l * 64 + d0;
#pragma unroll
          for (int j = 0; j < 8; ++j) g[l] += gf[j] * vr[j];
        }
    }
  }
  float ds[XB_L];
  if (p.L == 1) {
    ds[0] = pr[0] * (1.0f - pr[0]) * g[0] * p.scale;
  } else {
    float dot = 0.f;
#pragma unroll
    for (int l = 0; l < XB_L; ++l) dot += pr[l] * g[l];
#pragma unroll
    for (int l = 0; l < XB_L; ++l) ds[l] = pr[l] * (g[l] - dot) * p.scale;
  }
  uint16_t* orow = p.dq + ((long long)b * p.nq + qi) * p.lddq + h * 64;
  for (int d0 = 0; d0 < 64; d0 += 8) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int l = 0; l < XB_L; ++l)
      if (l < p.L) {
        const float* kr = ks + l * 64 + d0;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += ds[l] * kr[j];
      }
    const u32x4 o4 = {pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7])};
    *reinterpret_cast<u32x4*>(orow + d0) = o4;
  }
}

// ------------------------------------------------------------------------------------------------ local loss backward
// forward (elementwise.hip local_loss_kernel, reference loss.py:192-235): per sample and layer
//   term = -min_l [ max_n( mask[n] * blur3x3(mean_h P[h, n, l])[n] ) + (1 - seg[l]) ].
// The gradient is the 3x3 stencil of the blur around the arg-max pixel n* of the arg-min token l*, spread evenly over the heads:
//   dP[h, n* + off(tap), l*] += weight * (-1) * mask[n*] * gk[tap] / heads.
// One workgroup per (context token, sample) scores the token (head mean, blur, masked arg-max: stride-L reads of the sample's maps out
// of L2), then one workgroup per sample takes the arg-min token and writes the stencil.  arg-max / arg-min take the FIRST extremum, as
// torch.max / torch.min do on the CPU.  d_probs must be zero-initialised by the caller (other contributions may be accumulated into
// it afterwards).  (One workgroup per sample looping over the tokens: 184 us per call at 64 x 64 — a batch-1 attend-and-excite
// evaluation makes fifteen.)  scratch: fp32 [n_samples][seg_l][2] = (masked maximum, its pixel index as int bits).
UDT_DEVINL void ll_head_mean(const float* __restrict__ probs, float* amap, int b, int l, int heads, int n, int L) {
  for (int i = threadIdx.x; i < n; i += 256) {
    float a = 0.f;
    for (int hh = 0; hh < heads; ++hh) a += probs[(((long long)b * heads + hh) * n + i) * L + l];
    amap[i] = a / (float)heads;
  }
}

UDT_DEVINL float ll_blur(const float* amap, const float* __restrict__ gk, int y, int x, int size) {
  float acc = 0.f;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int yy = y + dy, xx = x + dx;
      if (yy >= 0 && yy < size && xx >= 0 && xx < size) acc += gk[(dy + 1) * 3 + dx + 1] * amap[yy * size + xx];
    }
  return acc;
}

UDT_DEVINL void ll_wave_argmax(float& mx, int& mi) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(mx, off);
    const int oi = __shfl_xor(mi, off);
    if (ov > mx || (ov == mx && oi < mi)) { mx = ov; mi = oi; }
  }
}

__global__ void __launch_bounds__(256) local_loss_token_kernel(const float* __restrict__ probs, const float* __restrict__ mask,
                                                               const float* __restrict__ gk, float* __restrict__ scratch, int heads,
                                                               int size, int L, int seg_l, int Hm, int Wm, int mask_batch) {
  extern __shared__ __attribute__((aligned(16))) float lbsm[];
  float* amap = lbsm;                         // [size * size]
  float* redv = lbsm + size * size;           // [4] wave maxima
  int* redi = reinterpret_cast<int*>(redv + 4);   // [4] their pixel indices
  const int l = blockIdx.x, b = blockIdx.y, bm = b % mask_batch, t = threadIdx.x, n = size * size;
  ll_head_mean(probs, amap, b, l, heads, n, L);
  __syncthreads();
  float mx = -INFINITY;
  int mi = 0x7fffffff;
  for (int i = t; i < n; i += 256) {
    const int y = i / size, x = i - y * size;
    const float acc = ll_blur(amap, gk, y, x, size);
    const int my = (int)(((long long)y * Hm) / size), mxx = (int)(((long long)x * Wm) / size);
    const float val = mask[((long long)bm * Hm + my) * Wm + mxx] * acc;
    if (val > mx) { mx = val; mi = i; }      // (i ascends per thread: the first maximum of the thread's pixels)
  }
  ll_wave_argmax(mx, mi);
  if ((t & 63) == 0) { redv[t >> 6] = mx; redi[t >> 6] = mi; }
  __syncthreads();
  if (t == 0) {
    float m4 = redv[0];
    int i4 = redi[0];
    for (int w = 1; w < 4; ++w)
      if (redv[w] > m4 || (redv[w] == m4 && redi[w] < i4)) { m4 = redv[w]; i4 = redi[w]; }
    float* dst = scratch + ((long long)b * seg_l + l) * 2;
    dst[0] = m4;
    reinterpret_cast<int*>(dst)[1] = i4 == 0x7fffffff ? 0 : i4;
  }
}

__global__ void __launch_bounds__(256) local_loss_finish_kernel(const float* __restrict__ scratch, const float* __restrict__ mask,
                                                                const float* __restrict__ seg, const float* __restrict__ gk,
                                                                float* __restrict__ d_probs, float* __restrict__ loss, int heads, int size,
                                                                int L, int seg_l, int Hm, int Wm, int mask_batch, float weight) {
  const int b = blockIdx.x, bm = b % mask_batch, t = threadIdx.x, n = size * size;
  float best = INFINITY;                       // (every thread walks the <= 16 tokens the same way)
  int ls = 0, ns = 0;
  for (int l = 0; l < seg_l; ++l) {
    const float* src = scratch + ((long long)b * seg_l + l) * 2;
    const float pl = src[0] + (1.0f - seg[(long long)bm * seg_l + l]);
    if (pl < best) { best = pl; ls = l; ns = reinterpret_cast<const int*>(src)[1]; }
  }
  if (t == 0 && loss) loss[b] += -best;
  const int ys = ns / size, xs = ns - ys * size;
  const int my = (int)(((long long)ys * Hm) / size), mxx = (int)(((long long)xs * Wm) / size);
  const float coef = -weight * mask[((long long)bm * Hm + my) * Wm + mxx] / (float)heads;
  for (int i = t; i < 9 * heads; i += 256) {
    const int hh = i / 9, tap = i - hh * 9;
    const int yy = ys + tap / 3 - 1, xx = xs + tap % 3 - 1;
    if (yy >= 0 && yy < size && xx >= 0 && xx < size)
      d_probs[(((long long)b * heads + hh) * n + yy * size + xx) * L + ls] += coef * gk[tap];
  }
}

// ------------------------------------------------------------------------------------------------ LayerNorm backward
// y = xhat * gamma + beta, xhat = (x - mean) * rstd over the C channels of a row:
//   g = dy * gamma;  dx = rstd * (g - mean(g) - xhat * mean(g * xhat))  (+ add: the gradient arriving over the residual path).
// One wave per row; NCH 16-byte chunks per lane (C <= NCH * 512), the row stays in registers.
template <int NCH>
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                                            const float* __restrict__ gamma, const uint16_t* __restrict__ add,
                                                            uint16_t* __restrict__ dx, long long rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int c8 = C >> 3;
  float xv[NCH][8], gv[NCH][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + i * 64;
    if (ch < c8) {
      const u32x4 u = *reinterpret_cast<const u32x4*>(x + row * C + ch * 8);
      const u32x4 w = *reinterpret_cast<const u32x4*>(dy + row * C + ch * 8);
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + ch * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + ch * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xv[i][2 * j] = bf16_lo(u[j]);
        xv[i][2 * j + 1] = bf16_hi(u[j]);
        gv[i][2 * j] = bf16_lo(w[j]);
        gv[i][2 * j + 1] = bf16_hi(w[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { gv[i][j] *= g0[j]; gv[i][4 + j] *= g1[j]; }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) xv[i][j] = gv[i][j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s += xv[i][j];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
    if (lane + i * 64 < c8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = xv[i][j] - mean; q += d * d; }
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
  const float rstd = rsqrtf(q / (float)C + eps);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
    if (lane + i * 64 < c8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xv[i][j] = (xv[i][j] - mean) * rstd;         // xhat
        s1 += gv[i][j];
        s2 += gv[i][j] * xv[i][j];
      }
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
  s1 /= (float)C;
  s2 /= (float)C;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + i * 64;
    if (ch < c8) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = rstd * (gv[i][j] - s1 - xv[i][j] * s2);
      if (add) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(add + row * C + ch * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[2 * j] += bf16_lo(a[j]); o[2 * j + 1] += bf16_hi(a[j]); }
      }
      const u32x4 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
      *reinterpret_cast<u32x4*>(dx + row * C + ch * 8) = pk;
    }
  }
}

// ------------------------------------------------------------------------------------------------ GroupNorm (+ SiLU) backward
// y = act(xhat * gamma_c + beta_c), xhat = (x - mean) * rstd over the (pixels x channels-of-the-group) of one sample; act = SiLU or
// identity.  dz = dy * act'(xhat gamma + beta);  g = dz * gamma;  dx = rstd * (g - mean(g) - xhat * mean(g xhat)) (+ add).
// One workgroup per (sample, group): four passes over the group's elements (sum; squared deviations; the two gradient sums; the
// result), the later ones out of L2.  Channel-last layout: a pixel's cpg channels are contiguous, cpg is even: dword accesses.
UDT_DEVINL float block_sum_256(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(256) gn_bwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const uint16_t* __restrict__ add, uint16_t* __restrict__ dx, int HW, int C,
                                                     int groups, float eps, int silu) {
  __shared__ float red[4];
  const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
  const int cpg = C / groups, hp = cpg >> 1;
  const long long base = (long long)b * HW * C + g * cpg;
  const int n2 = HW * hp;                              // dwords of the group
  const float inv_n = 1.0f / (float)(HW * cpg);
  auto addr = [&](int e) { const int px = e / hp, j = e - px * hp; return base + (long long)px * C + 2 * j; };
  float s = 0.f;
  for (int e = threadIdx.x; e < n2; e += 256) {
    const uint32_t u = *reinterpret_cast<const uint32_t*>(x + addr(e));
    s += bf16_lo(u) + bf16_hi(u);
  }
  const float mean = block_sum_256(s, red) * inv_n;
  float q = 0.f;
  for (int e = threadIdx.x; e < n2; e += 256) {
    const uint32_t u = *reinterpret_cast<const uint32_t*>(x + addr(e));
    const float a = bf16_lo(u) - mean, c2 = bf16_hi(u) - mean;
    q += a * a + c2 * c2;
  }
  const float rstd = rsqrtf(block_sum_256(q, red) * inv_n + eps);
  auto dz = [&](float xh, float dyv, float ga, float be) {
    if (!silu) return dyv;
    const float y0 = xh * ga + be;
    const float sg = 1.0f / (1.0f + __expf(-y0));
    return dyv * sg * (1.0f + y0 * (1.0f - sg));
  };
  float s1 = 0.f, s2 = 0.f;
  for (int e = threadIdx.x; e < n2; e += 256) {
    const int j = e % hp;
    const long long a = addr(e);
    const uint32_t u = *reinterpret_cast<const uint32_t*>(x + a);
    const uint32_t w = *reinterpret_cast<const uint32_t*>(dy + a);
    const float g0 = gamma[g * cpg + 2 * j], g1 = gamma[g * cpg + 2 * j + 1];
    const float b0 = beta[g * cpg + 2 * j], b1 = beta[g * cpg + 2 * j + 1];
    const float x0 = (bf16_lo(u) - mean) * rstd, x1 = (bf16_hi(u) - mean) * rstd;
    const float t0 = dz(x0, bf16_lo(w), g0, b0) * g0, t1 = dz(x1, bf16_hi(w), g1, b1) * g1;
    s1 += t0 + t1;
    s2 += t0 * x0 + t1 * x1;
  }
  s1 = block_sum_256(s1, red) * inv_n;
  s2 = block_sum_256(s2, red) * inv_n;
  for (int e = threadIdx.x; e < n2; e += 256) {
    const int j = e % hp;
    const long long a = addr(e);
    const uint32_t u = *reinterpret_cast<const uint32_t*>(x + a);
    const uint32_t w = *reinterpret_cast<const uint32_t*>(dy + a);
    const float g0 = gamma[g * cpg + 2 * j], g1 = gamma[g * cpg + 2 * j + 1];
    const float b0 = beta[g * cpg + 2 * j], b1 = beta[g * cpg + 2 * j + 1];
    const float x0 = (bf16_lo(u) - mean) * rstd, x1 = (bf16_hi(u) - mean) * rstd;
    float o0 = rstd * (dz(x0, bf16_lo(w), g0, b0) * g0 - s1 - x0 * s2);
    float o1 = rstd * (dz(x1, bf16_hi(w), g1, b1) * g1 - s1 - x1 * s2);
    if (add) {
      const uint32_t r = *reinterpret_cast<const uint32_t*>(add + a);
      o0 += bf16_lo(r);
      o1 += bf16_hi(r);
    }
    *reinterpret_cast<uint32_t*>(dx + a) = pack_bf16x2(o0, o1);
  }
}

// The same gradient at the parallelism of the forward's GroupNorm (norm.hip gn_stats_kernel / gn_apply_kernel): the one-workgroup-per-
// (sample, group) kernel above runs 32 workgroups for a batch-1 attend-and-excite call and reads 20-byte pieces of 640-byte pixel
// rows (96 us per call, 5.7 ms of a 24 ms evaluation).  Three launches over whole pixel rows in 16-byte pieces:
//   udt_gn_stats (forward's)   chunk partials of sum x, sum x^2                          -> part [B][nchunks][G][2]
//   gn_bwd_sums_kernel         chunk partials of sum t, sum t xhat, t = dz gamma          -> part2 [B][nchunks][G][2]
//   gn_bwd_apply_kernel        dx = rstd (t - mean(t) - xhat mean(t xhat)) (+ add), every workgroup reduces the partials itself
// (double accumulation across chunks, as the forward's apply does).
UDT_DEVINL float gn_dz(int silu, float xh, float dyv, float ga, float be) {
  if (!silu) return dyv;
  const float y0 = xh * ga + be;
  const float sg = 1.0f / (1.0f + __expf(-y0));
  return dyv * sg * (1.0f + y0 * (1.0f - sg));
}

// out0[g], out1[g] (g < G) = the two chunk-partial sums of group g over the sample's chunks; red: [2][256] doubles.  All 256 threads call.
UDT_DEVINL void gn_reduce_partials(const float* __restrict__ partials, int b, int nchunks, int G, double* red, double& a_out, double& q_out) {
  const int t = threadIdx.x;
  const int lanes = 256 / G;
  const int g = t % G, ln = t / G;
  double a = 0.0, q = 0.0;
  if (ln < lanes)
    for (int k = ln; k < nchunks; k += lanes) {
      const float* src = partials + (((long long)b * nchunks + k) * G + g) * 2;
      a += (double)src[0];
      q += (double)src[1];
    }
  __syncthreads();
  red[t] = a;
  red[256 + t] = q;
  __syncthreads();
  a = 0.0; q = 0.0;
  if (t < G)
    for (int l = 0; l < lanes; ++l) { a += red[l * G + t]; q += red[256 + l * G + t]; }
  a_out = a;
  q_out = q;
}

// grid (nchunks, B); thread layout of gn_stats_kernel: R row-groups x c8 channel chunks of 8
__global__ void __launch_bounds__(256) gn_bwd_sums_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                                          const float* __restrict__ partials, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ part2, long long HW, int C,
                                                          int G, int nchunks, float eps, int silu) {
  extern __shared__ __attribute__((aligned(16))) double gbsm[];
  double* red = gbsm;                                          // [512]
  float* gm = reinterpret_cast<float*>(gbsm + 512);            // [G] mean
  float* gr = gm + G;                                          // [G] rstd
  float* b1 = gr + G;                                          // [R][C] sum t
  const int t = threadIdx.x, chunk = blockIdx.x, b = blockIdx.y;
  const int c8 = C >> 3, cpg = C / G;
  double a, q;
  gn_reduce_partials(partials, b, nchunks, G, red, a, q);
  if (t < G) {
    const double n = (double)HW * (double)cpg;
    const double mean = a / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    gm[t] = (float)mean;
    gr[t] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const long long rows_per_chunk = (HW + nchunks - 1) / nchunks;
  const long long p0 = chunk * rows_per_chunk;
  long long p1 = p0 + rows_per_chunk;
  if (p1 > HW) p1 = HW;
  int R, rg, cc0, cstep;
  if (c8 <= 256) {
    R = 256 / c8; rg = t / c8; cc0 = t - rg * c8; cstep = c8;
    if (rg >= R) cc0 = c8;
  } else {
    R = 1; rg = 0; cc0 = t; cstep = 256;
  }
  float* b2 = b1 + R * C;                                      // [R][C] sum t xhat
  for (int cc = cc0; cc < c8; cc += cstep) {
    float ga[8], be[8], mu[8], rs[8], s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cc * 8 + j, g = c / cpg;
      ga[j] = gamma[c]; be[j] = beta[c]; mu[j] = gm[g]; rs[j] = gr[g];
      s1[j] = s2[j] = 0.f;
    }
    const uint16_t* xb = x + (long long)b * HW * C + cc * 8;
    const uint16_t* db = dy + (long long)b * HW * C + cc * 8;
    for (long long pix = p0 + rg; pix < p1; pix += R) {
      const u32x4 xv = *reinterpret_cast<const u32x4*>(xb + pix * C);
      const u32x4 dv = *reinterpret_cast<const u32x4*>(db + pix * C);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x0 = (bf16_lo(xv[j]) - mu[2 * j]) * rs[2 * j], x1 = (bf16_hi(xv[j]) - mu[2 * j + 1]) * rs[2 * j + 1];
        const float t0 = gn_dz(silu, x0, bf16_lo(dv[j]), ga[2 * j], be[2 * j]) * ga[2 * j];
        const float t1 = gn_dz(silu, x1, bf16_hi(dv[j]), ga[2 * j + 1], be[2 * j + 1]) * ga[2 * j + 1];
        s1[2 * j] += t0; s2[2 * j] += t0 * x0;
        s1[2 * j + 1] += t1; s2[2 * j + 1] += t1 * x1;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      b1[rg * C + cc * 8 + j] = s1[j];
      b2[rg * C + cc * 8 + j] = s2[j];
    }
  }
  __syncthreads();
  if (t < G) {
    float sa = 0.f, sb = 0.f;
    for (int r = 0; r < R; ++r)
      for (int c = t * cpg; c < (t + 1) * cpg; ++c) {
        sa += b1[r * C + c];
        sb += b2[r * C + c];
      }
    float* dst = part2 + (((long long)b * nchunks + chunk) * G + t) * 2;
    dst[0] = sa;
    dst[1] = sb;
  }
}

// grid (blocks per sample, B): spans of chunks_per_wg 16-byte pieces of one sample
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                                           const uint16_t* __restrict__ add, uint16_t* __restrict__ dx,
                                                           const float* __restrict__ partials, const float* __restrict__ part2,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, long long HW,
                                                           int C, int G, int nchunks, float eps, int silu, long long chunks_per_wg) {
  extern __shared__ __attribute__((aligned(16))) double gbsm[];
  double* red = gbsm;                                          // [512]
  float* gs1 = reinterpret_cast<float*>(gbsm + 512);           // [G] mean(t)
  float* gs2 = gs1 + G;                                        // [G] mean(t xhat)
  float* gm = gs2 + G;
  float* gr = gm + G;
  float* tA = gr + G;                                          // [C] rstd of the channel's group
  float* tB = tA + C;                                          // [C] -mean rstd
  float* tG = tB + C;                                          // [C] gamma
  float* tE = tG + C;                                          // [C] beta
  const int t = threadIdx.x, b = blockIdx.y;
  const int c8 = C >> 3, cpg = C / G;
  const double n = (double)HW * (double)cpg;
  double a, q;
  gn_reduce_partials(partials, b, nchunks, G, red, a, q);
  if (t < G) {
    const double mean = a / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    gm[t] = (float)mean;
    gr[t] = (float)(1.0 / sqrt(var + (double)eps));
  }
  gn_reduce_partials(part2, b, nchunks, G, red, a, q);
  if (t < G) {
    gs1[t] = (float)(a / n);
    gs2[t] = (float)(q / n);
  }
  __syncthreads();
  for (int c = t; c < C; c += 256) {
    const int g = c / cpg;
    tA[c] = gr[g];
    tB[c] = -gm[g] * gr[g];
    tG[c] = gamma[c];
    tE[c] = beta[c];
  }
  __syncthreads();
  const long long total = HW * c8;
  const long long begin = (long long)blockIdx.x * chunks_per_wg;
  long long end = begin + chunks_per_wg;
  if (end > total) end = total;
  const long long sb = (long long)b * HW * C;
  for (long long i = begin + t; i < end; i += 256) {
    const int cc = (int)(i % c8);
    const u32x4 xv = *reinterpret_cast<const u32x4*>(x + sb + i * 8);
    const u32x4 dv = *reinterpret_cast<const u32x4*>(dy + sb + i * 8);
    u32x4 av = {0u, 0u, 0u, 0u};
    if (add) av = *reinterpret_cast<const u32x4*>(add + sb + i * 8);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = cc * 8 + 2 * j;
      const int g = c / cpg;                                    // (cpg is even: both channels of the pair lie in group g)
      const float m1 = gs1[g], m2 = gs2[g];
      const float x0 = bf16_lo(xv[j]) * tA[c] + tB[c], x1 = bf16_hi(xv[j]) * tA[c + 1] + tB[c + 1];
      const float t0 = gn_dz(silu, x0, bf16_lo(dv[j]), tG[c], tE[c]) * tG[c];
      const float t1 = gn_dz(silu, x1, bf16_hi(dv[j]), tG[c + 1], tE[c + 1]) * tG[c + 1];
      const float o0 = tA[c] * (t0 - m1 - x0 * m2) + bf16_lo(av[j]);
      const float o1 = tA[c + 1] * (t1 - m1 - x1 * m2) + bf16_hi(av[j]);
      o[j] = pack_bf16x2(o0, o1);
    }
    *reinterpret_cast<u32x4*>(dx + sb + i * 8) = o;
  }
}

// ------------------------------------------------------------------------------------------------ GEGLU on stored pre-activations
// ag [rows][2 * inner]: columns [0, inner) = x, [inner, 2 inner) = gate (reference attention.py:44-52: x, gate = proj(x).chunk(2);
// out = x * gelu(gate), exact-erf GELU).  d x = dy * gelu(gate);  d gate = dy * x * (Phi(gate) + gate * phi(gate)).
UDT_DEVINL float erf_as(float x) {                     // Abramowitz-Stegun 7.1.26, as gelu_erf_f
  const float z = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  float poly = 1.061405429f;
  poly = poly * t - 1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t - 0.284496736f;
  poly = poly * t + 0.254829592f;
  poly = poly * t;
  return __builtin_copysignf(1.0f - poly * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f), x);
}

template <bool BWD>
__global__ void __launch_bounds__(256) geglu_kernel(const uint16_t* __restrict__ ag, const uint16_t* __restrict__ dy,
                                                    uint16_t* __restrict__ out, long long n8, int inner8) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const long long row = i / inner8;
  const int c = (int)(i - row * inner8) * 8;
  const long long inner = (long long)inner8 * 8;
  const uint16_t* pa = ag + row * 2 * inner + c;
  const u32x4 xv = *reinterpret_cast<const u32x4*>(pa);
  const u32x4 gvv = *reinterpret_cast<const u32x4*>(pa + inner);
  float xf[8], gf[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    xf[2 * j] = bf16_lo(xv[j]); xf[2 * j + 1] = bf16_hi(xv[j]);
    gf[2 * j] = bf16_lo(gvv[j]); gf[2 * j + 1] = bf16_hi(gvv[j]);
  }
  if constexpr (!BWD) {
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = xf[j] * gelu_erf_f(gf[j]);
    const u32x4 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
    *reinterpret_cast<u32x4*>(out + row * inner + c) = pk;
  } else {
    const u32x4 dv = *reinterpret_cast<const u32x4*>(dy + row * inner + c);
    float df[8], ox[8], og[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { df[2 * j] = bf16_lo(dv[j]); df[2 * j + 1] = bf16_hi(dv[j]); }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float phi_c = 0.5f * (1.0f + erf_as(gf[j] * 0.70710678118654752f));
      const float pdf = 0.3989422804014327f * __expf(-0.5f * gf[j] * gf[j]);
      ox[j] = df[j] * gf[j] * phi_c;
      og[j] = df[j] * xf[j] * (phi_c + gf[j] * pdf);
    }
    uint16_t* po = out + row * 2 * inner + c;
    const u32x4 px = {pack_bf16x2(ox[0], ox[1]), pack_bf16x2(ox[2], ox[3]), pack_bf16x2(ox[4], ox[5]), pack_bf16x2(ox[6], ox[7])};
    const u32x4 pg = {pack_bf16x2(og[0], og[1]), pack_bf16x2(og[2], og[3]), pack_bf16x2(og[4], og[5]), pack_bf16x2(og[6], og[7])};
    *reinterpret_cast<u32x4*>(po) = px;
    *reinterpret_cast<u32x4*>(po + inner) = pg;
  }
}

// nearest x2 upsampling backward (Upsample.forward, openaimodel.py:99-101: F.interpolate(scale_factor=2, mode="nearest")): every input
// pixel fed a 2 x 2 block of the upsampled map, its gradient is the sum of the block.  dy [B, 2H, 2W, C] -> dx [B, H, W, C], 8 channels
// per thread, fp32 sum, one bf16 rounding.
__global__ void __launch_bounds__(256) sum2x2_kernel(const uint16_t* __restrict__ dy, uint16_t* __restrict__ dx, int H, int W, int c8,
                                                     long long n8) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const int c = (int)(i % c8);
  long long px = i / c8;
  const int x = (int)(px % W);
  px /= W;
  const int y = (int)(px % H);
  const long long b = px / H;
  const long long row = (long long)c8 * 8;
  const uint16_t* src = dy + (((b * 2 * H + 2 * y) * 2 * W) + 2 * x) * row + c * 8;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int dyy = 0; dyy < 2; ++dyy)
#pragma unroll
    for (int dxx = 0; dxx < 2; ++dxx) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(src + ((long long)dyy * 2 * W + dxx) * row);
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[2 * j] += bf16_lo(v[j]); acc[2 * j + 1] += bf16_hi(v[j]); }
    }
  const u32x4 pk = {pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7])};
  *reinterpret_cast<u32x4*>(dx + i * 8) = pk;
}

// Context tokens minus their mean over the L tokens of a sample: out[b, l, :] = bf16(x[b, l, :] - mean_l x[b, l, :]).  The text
// cross-attention's softmax over the tokens is invariant under a common shift of the keys (every logit moves by q . kbar), and so is its
// backward under a common shift of the values (dS_l = P_l (g_l - sum_j P_j g_j)): projecting the CENTRED context keeps in bf16 what the
// softmax and its gradient actually see — the differences between the tokens.  The label embeddings of one string are nearly equal
// vectors (12 positions of one transformer encoder), and dq = scale sum_l dS_l K_l with sum_l dS_l = 0 cancels their common part:
// with K rounded to bf16 un-centred, the attend-and-excite gradient carried 7 % of rounding noise (profiles/r06_aae_debug.txt).
__global__ void __launch_bounds__(256) center_tokens_kernel(const float* __restrict__ x, uint16_t* __restrict__ out, int L, int D) {
  const int b = blockIdx.y;
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  const float* xb = x + (long long)b * L * D + d;
  float m = 0.f;
  for (int l = 0; l < L; ++l) m += xb[(long long)l * D];
  m /= (float)L;
  uint16_t* ob = out + (long long)b * L * D + d;
  for (int l = 0; l < L; ++l) ob[(long long)l * D] = (uint16_t)(pack_bf16x2(xb[(long long)l * D] - m, 0.f) & 0xffffu);
}

// x += a * y (fp32): the attend-and-excite update x <- x - alpha * grad (reference sampling.py:247) on the sampler's fp32 state
__global__ void __launch_bounds__(256) axpy_f32_kernel(float* __restrict__ x, const float* __restrict__ y, float a, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] = __builtin_fmaf(a, y[i], x[i]);
}

// ================================================================================================ training step (second half)
// The reference trains the text cross-attention only (configs/train/textdesign_sd_2.yaml:4-6 opt_keys t_attn, t_norm: 75.9 M of the
// UNet's parameters): FullLoss.__call__ (loss.py:131-176) = weighted eps-prediction loss + lambda * get_local_loss (loss.py:237-286).
// Parameter gradients need, beyond the dX reverse pass above: dW = dY^T X products (the forward GEMM on transposed operands:
// transpose_bf16_kernel), the context-side gradients of the text cross-attention (dK, dV), LayerNorm's d gamma / d beta, bias column
// sums, the two losses' seeds, and the AdamW update.

// in [R][ld] bf16 (first C columns) -> out [C][Rp] bf16, Rp >= R a multiple of 64; columns R .. Rp - 1 are zero-filled
__global__ void __launch_bounds__(256) transpose_bf16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int R, int C,
                                                            int ld, int Rp) {
  __shared__ uint16_t tile[64][66];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < R && c0 + c < C) ? in[(long long)(r0 + r) * ld + c0 + c] : (uint16_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (c0 + c < C && r0 + r < Rp) out[(long long)(c0 + c) * Rp + r0 + r] = tile[r][c];
  }
}

// out[n] = sum_p in[p][n]  (fp32; fixed order: deterministic).  A workgroup = 16 columns x 16 lanes over p (independent loads in
// flight), combined through LDS in lane order.
__global__ void __launch_bounds__(256) reduce_rows_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int P, long long n,
                                                             int accumulate) {
  __shared__ float red[16][17];
  const int col = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const long long i = (long long)blockIdx.x * 16 + col;
  float a = 0.f;
  if (i < n) {
    int p = pl;
    for (; p + 48 < P; p += 64) {
      const float v0 = in[(long long)p * n + i], v1 = in[(long long)(p + 16) * n + i];
      const float v2 = in[(long long)(p + 32) * n + i], v3 = in[(long long)(p + 48) * n + i];
      a += (v0 + v1) + (v2 + v3);
    }
    for (; p < P; p += 16) a += in[(long long)p * n + i];
  }
  red[pl][col] = a;
  __syncthreads();
  if (pl == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][col];
    out[i] = accumulate ? out[i] + t : t;
  }
}

// column partial sums of bf16 rows: part[wg * 4 + wave][c] = sum over that wave's rows of x[r][c]   (bias gradients)
__global__ void __launch_bounds__(256) colsum_partial_kernel(const uint16_t* __restrict__ x, float* __restrict__ part, long long rows, int C,
                                                             int rows_per_wg) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long r0 = (long long)blockIdx.x * rows_per_wg;
  long long r1 = r0 + rows_per_wg;
  if (r1 > rows) r1 = rows;
  float* dst = part + ((long long)blockIdx.x * 4 + wave) * C;
  for (int c = lane * 2; c < C; c += 128) {
    float a0 = 0.f, a1 = 0.f;
    long long r = r0 + wave;
    for (; r + 12 < r1; r += 16) {                              // four independent loads per lane
      const uint32_t u0 = *reinterpret_cast<const uint32_t*>(x + r * C + c);
      const uint32_t u1 = *reinterpret_cast<const uint32_t*>(x + (r + 4) * C + c);
      const uint32_t u2 = *reinterpret_cast<const uint32_t*>(x + (r + 8) * C + c);
      const uint32_t u3 = *reinterpret_cast<const uint32_t*>(x + (r + 12) * C + c);
      a0 += (bf16_lo(u0) + bf16_lo(u1)) + (bf16_lo(u2) + bf16_lo(u3));
      a1 += (bf16_hi(u0) + bf16_hi(u1)) + (bf16_hi(u2) + bf16_hi(u3));
    }
    for (; r < r1; r += 4) {
      const uint32_t u = *reinterpret_cast<const uint32_t*>(x + r * C + c);
      a0 += bf16_lo(u);
      a1 += bf16_hi(u);
    }
    dst[c] = a0;
    dst[c + 1] = a1;
  }
}

// Weight gradient of a linear layer: dW fp32 [N][K] = sum_r dY[r][n] X[r][k]  (dY bf16 [R][N], X bf16 [R][K], row-major — the layout
// the reverse pass holds them in).  The contraction index is the ROW index of both operands, so both MFMA operands are transposed
// tiles: a workgroup (2 x 2 waves, 128 x 128 outputs, 64 x 64 per wave) stages 32 rows of both per step and writes them to LDS
// TRANSPOSED ([column][row], 16-bit stores, rows xor-swizzled in groups of eight against bank conflicts), from where the fragments are
// plain 16-byte reads; the next step's global loads are in flight during the MFMAs.  grid (tiles_k, tiles_n, S): the rows are cut
// into S ranges whose fp32 partial tiles go to `part` [S][N][K] and are added in range order by reduce_rows_f32_kernel (S = 1:
// straight to dW).  Replaces transpose x 2 + the forward GEMM with fp32 output (1280 launches of 8 us + 132 us per large one).
constexpr int WG_RS = 32;          // rows per step
constexpr int WG_P = 32;           // LDS pitch (elements) of a transposed row: 64 B, no padding — the swizzle spreads the banks

UDT_DEVINL int wg_pos(int col, int r) { return col * WG_P + ((((r >> 3) ^ (col >> 3)) & 3) << 3) + (r & 7); }

__global__ void __launch_bounds__(256) wgrad_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ x, float* __restrict__ out,
                                                    int R, int N, int K, int ldy, int ldx, int rows_per_split) {
  __shared__ __attribute__((aligned(16))) uint16_t yt[128 * WG_P];
  __shared__ __attribute__((aligned(16))) uint16_t xt[128 * WG_P];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int k0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
  const int r_begin = blockIdx.z * rows_per_split;
  const int r_end = min(R, r_begin + rows_per_split);
  const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;
  // staging: thread = (row & 7, chunk of 8 columns, row bit 3), piece i = 0, 1 adds 16 rows: a wave's 16-bit LDS stores cover
  // 8 rows x 8 chunks = 4 swizzle slots x 4 row pairs (two lanes per bank where one chunk x 64 rows would put sixteen)
  u32x4 gy[2], gx[2];
  auto gload = [&](int r0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = r0 + (tid & 7) + 8 * ((tid >> 7) + 2 * i), ch = (tid >> 3) & 15;
      const bool rok = row < r_end;
      u32x4 a = {0u, 0u, 0u, 0u}, b = {0u, 0u, 0u, 0u};
      if (rok && n0 + ch * 8 < N) a = *reinterpret_cast<const u32x4*>(dy + (long long)row * ldy + n0 + ch * 8);
      if (rok && k0 + ch * 8 < K) b = *reinterpret_cast<const u32x4*>(x + (long long)row * ldx + k0 + ch * 8);
      gy[i] = a;
      gx[i] = b;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (tid & 7) + 8 * ((tid >> 7) + 2 * i), ch = (tid >> 3) & 15;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        yt[wg_pos(ch * 8 + 2 * j, row)] = (uint16_t)(gy[i][j] & 0xffffu);
        yt[wg_pos(ch * 8 + 2 * j + 1, row)] = (uint16_t)(gy[i][j] >> 16);
        xt[wg_pos(ch * 8 + 2 * j, row)] = (uint16_t)(gx[i][j] & 0xffffu);
        xt[wg_pos(ch * 8 + 2 * j + 1, row)] = (uint16_t)(gx[i][j] >> 16);
      }
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  gload(r_begin);
  for (int r0 = r_begin; r0 < r_end; r0 += WG_RS) {
    __syncthreads();
    lstore();
    __syncthreads();
    gload(r0 + WG_RS);                                          // (rows >= r_end load nothing)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t af[2], bf[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int col = wn + a * 32 + l31;
        af[a] = *reinterpret_cast<const bf16x8_t*>(yt + wg_pos(col, ks * 16 + hi * 8));
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int col = wk + b * 32 + l31;
        bf[b] = *reinterpret_cast<const bf16x8_t*>(xt + wg_pos(col, ks * 16 + hi * 8));
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = mfma32(af[a], bf[b], acc[a][b]);
    }
  }
  float* dst = out + (long long)blockIdx.z * N * K;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int kk = k0 + wk + b * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nn = n0 + wn + a * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
        if (nn < N && kk < K) dst[(long long)nn * K + kk] = acc[a][b][r];
      }
    }
}

// LayerNorm parameter gradients, partial: part[wg * 4 + wave][0][c] = sum_r dy[r][c] * xhat[r][c], [1][c] = sum_r dy[r][c] over the
// wave's rows (one wave per row at a time: row statistics by shuffles, as layernorm_bwd_kernel)
template <int NCH>
__global__ void __launch_bounds__(256) ln_param_grad_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                                            float* __restrict__ part, long long rows, int C, float eps, int rows_per_wg) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c8 = C >> 3;
  float gg[NCH][8], gb[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) gg[i][j] = gb[i][j] = 0.f;
  const long long r0 = (long long)blockIdx.x * rows_per_wg;
  for (long long row = r0 + wave; row < r0 + rows_per_wg && row < rows; row += 4) {
    float xv[NCH][8], dv[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = lane + i * 64;
      if (ch < c8) {
        const u32x4 u = *reinterpret_cast<const u32x4*>(x + row * C + ch * 8);
        const u32x4 w = *reinterpret_cast<const u32x4*>(dy + row * C + ch * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xv[i][2 * j] = bf16_lo(u[j]); xv[i][2 * j + 1] = bf16_hi(u[j]);
          dv[i][2 * j] = bf16_lo(w[j]); dv[i][2 * j + 1] = bf16_hi(w[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[i][j] = dv[i][j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += xv[i][j];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (lane + i * 64 < c8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = xv[i][j] - mean; q += d * d; }
      }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = rsqrtf(q / (float)C + eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        gg[i][j] += dv[i][j] * (xv[i][j] - mean) * rstd;
        gb[i][j] += dv[i][j];
      }
  }
  float* dst = part + ((long long)blockIdx.x * 4 + wave) * 2 * C;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + i * 64;
    if (ch < c8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { dst[ch * 8 + j] = gg[i][j]; dst[C + ch * 8 + j] = gb[i][j]; }
    }
  }
}

// Text cross-attention, context side: dV[b, l, h d] = sum_n P[n][l] dO[n][d],  dK[b, l, h d] = scale sum_n dS[n][l] q[n][d]  with dS as
// in xattn_bwd_kernel.  One workgroup per (head, sample, range of XKV_QT queries): tiles of 64 queries through LDS (their P rows, dS
// rows, dO and q rows as fp32), thread t owns the outputs (l, d) = (t / 64 + 4 i, t % 64), i = 0 .. 3 (L <= 16) and writes its fp32
// partial sums; xattn_bwd_kv_reduce_kernel adds the ranges in ascending order (fixed summation order, no atomics).
// (One workgroup per (head, sample) over ALL queries was 1.3 ms per call at 4096 queries: 20 workgroups on 256 CUs.)
constexpr int XKV_QT = 128;

__global__ void __launch_bounds__(256) xattn_bwd_kv_kernel(const XattnBwdParams p, const uint16_t* __restrict__ q, int ldq,
                                                           float* __restrict__ part, int batch) {
  __shared__ float vs[XB_L * 64];
  __shared__ float pt[64 * XB_L];
  __shared__ float st[64 * XB_L];
  __shared__ float gt[64 * 65];
  __shared__ float qt[64 * 65];
  const int h = blockIdx.x, b = blockIdx.y, z = blockIdx.z, t = threadIdx.x;
  for (int i = t; i < XB_L * 64; i += 256) {
    const int l = i >> 6, d = i & 63;
    vs[i] = l < p.L ? bf16_bits_to_f32(p.v[((long long)b * p.L + l) * p.ldkv + h * 64 + d]) : 0.f;
  }
  float accv[4] = {0.f, 0.f, 0.f, 0.f}, acck[4] = {0.f, 0.f, 0.f, 0.f};
  const int d_own = t & 63, l_own = t >> 6;
  const int nend = min(p.nq, (z + 1) * XKV_QT);
  for (int n0 = z * XKV_QT; n0 < nend; n0 += 64) {
    __syncthreads();
    for (int i = t; i < 64 * 64; i += 256) {
      const int r = i >> 6, d = i & 63;
      const int n = n0 + r;
      const bool ok = n < p.nq;
      gt[r * 65 + d] = (ok && p.d_o) ? bf16_bits_to_f32(p.d_o[((long long)b * p.nq + n) * p.ldo + h * 64 + d]) : 0.f;
      qt[r * 65 + d] = ok ? bf16_bits_to_f32(q[((long long)b * p.nq + n) * ldq + h * 64 + d]) : 0.f;
    }
    __syncthreads();
    {
      // thread (r, lp): query r of the tile, context tokens 4 lp .. 4 lp + 3; the four lanes of a query are neighbours
      const int r = t >> 2, lp = t & 3, n = n0 + r;
      float pr[4], g[4];
      const long long prow = (((long long)b * p.heads + h) * p.nq + n) * p.L;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int l = lp * 4 + i;
        const bool ok = n < p.nq && l < p.L;
        pr[i] = ok ? p.probs[prow + l] : 0.f;
        g[i] = (ok && p.d_probs) ? p.d_probs[prow + l] : 0.f;
      }
      if (p.d_o) {
        for (int d = 0; d < 64; ++d) {
          const float gv = gt[r * 65 + d];
#pragma unroll
          for (int i = 0; i < 4; ++i) g[i] += gv * vs[(lp * 4 + i) * 64 + d];
        }
      }
      float dot = (pr[0] * g[0] + pr[1] * g[1]) + (pr[2] * g[2] + pr[3] * g[3]);
      dot += __shfl_xor(dot, 1);
      dot += __shfl_xor(dot, 2);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        pt[r * XB_L + lp * 4 + i] = pr[i];
        st[r * XB_L + lp * 4 + i] = (p.L == 1 ? pr[i] * (1.0f - pr[i]) * g[i] : pr[i] * (g[i] - dot)) * p.scale;
      }
    }
    __syncthreads();
    for (int r = 0; r < 64; ++r) {
      const float gv = gt[r * 65 + d_own], qv = qt[r * 65 + d_own];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int l = l_own + 4 * i;
        accv[i] += pt[r * XB_L + l] * gv;
        acck[i] += st[r * XB_L + l] * qv;
      }
    }
  }
  const int Cc = p.heads * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int l = l_own + 4 * i;
    if (l < p.L) {
      float* dst = part + ((((long long)z * batch + b) * p.L + l) * Cc + h * 64 + d_own) * 2;
      dst[0] = acck[i];
      dst[1] = accv[i];
    }
  }
}

// dk, dv bf16 [B * L][lddkv] = the sum over the S query ranges of part fp32 [S][B * L][C][2]
__global__ void __launch_bounds__(256) xattn_bwd_kv_reduce_kernel(const float* __restrict__ part, int S, long long rows, int Cc,
                                                                  uint16_t* __restrict__ dk, uint16_t* __restrict__ dv, int lddkv) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * Cc) return;
  float a = 0.f, c = 0.f;
  for (int z = 0; z < S; ++z) {
    const float* src = part + ((long long)z * rows * Cc + i) * 2;
    a += src[0];
    c += src[1];
  }
  const long long r = i / Cc;
  const int col = (int)(i - r * Cc);
  dk[r * lddkv + col] = (uint16_t)(pack_bf16x2(a, 0.f) & 0xffffu);
  dv[r * lddkv + col] = (uint16_t)(pack_bf16x2(c, 0.f) & 0xffffu);
}

// get_local_loss (reference loss.py:237-286) and its gradient: per sample and layer
//   f = sum_l segm[l] (max_n((1 - seg[l, n]) A[l, n]) - max_n(seg[l, n] A[l, n])) / sum_l segm[l],   A = blur3x3(mean_h P).
// dP gets, per scored token, the blur stencil around the two arg-max pixels: + outside the character's segment, - inside.
// One workgroup per (scored token, sample); scratch fp32 [B][seg_l] receives the token's term, local_loss_seg_finish_kernel adds them
// in ascending token order.
__global__ void __launch_bounds__(256) local_loss_seg_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ segmap,
                                                                 const float* __restrict__ segm, const float* __restrict__ gk,
                                                                 float* __restrict__ d_probs, float* __restrict__ scratch, int heads,
                                                                 int size, int L, int seg_l, int Hs, int Ws, float weight) {
  extern __shared__ __attribute__((aligned(16))) float lssm[];
  float* amap = lssm;
  float* redv = lssm + size * size;            // [8] wave maxima (inside 0..3, outside 4..7)
  int* redi = reinterpret_cast<int*>(redv + 8);   // [8]
  const int l = blockIdx.x, b = blockIdx.y, t = threadIdx.x, n = size * size;
  const float sm = segm[(long long)b * seg_l + l];
  if (sm == 0.f) {                                             // (uniform over the workgroup)
    if (t == 0) scratch[(long long)b * seg_l + l] = 0.f;
    return;
  }
  float ssum = 0.f;
  for (int k = 0; k < seg_l; ++k) ssum += segm[(long long)b * seg_l + k];
  ll_head_mean(probs, amap, b, l, heads, n, L);
  __syncthreads();
  float mp = -INFINITY, mn = -INFINITY;
  int ip = 0x7fffffff, in_ = 0x7fffffff;
  for (int i = t; i < n; i += 256) {
    const int y = i / size, x = i - y * size;
    const float acc = ll_blur(amap, gk, y, x, size);
    const int sy = (int)(((long long)y * Hs) / size), sx = (int)(((long long)x * Ws) / size);
    const float sv = segmap[(((long long)b * seg_l + l) * Hs + sy) * Ws + sx];
    const float vp = sv * acc, vn = (1.0f - sv) * acc;
    if (vp > mp) { mp = vp; ip = i; }
    if (vn > mn) { mn = vn; in_ = i; }
  }
  ll_wave_argmax(mp, ip);
  ll_wave_argmax(mn, in_);
  if ((t & 63) == 0) { redv[t >> 6] = mp; redi[t >> 6] = ip; redv[4 + (t >> 6)] = mn; redi[4 + (t >> 6)] = in_; }
  __syncthreads();
  for (int w = 0; w < 4; ++w) {                              // (every thread combines the four waves the same way)
    if (redv[w] > mp || (redv[w] == mp && redi[w] < ip)) { mp = redv[w]; ip = redi[w]; }
    if (redv[4 + w] > mn || (redv[4 + w] == mn && redi[4 + w] < in_)) { mn = redv[4 + w]; in_ = redi[4 + w]; }
  }
  if (t == 0) scratch[(long long)b * seg_l + l] = sm * (mn - mp);
  const float cbase = weight * sm / ssum / (float)heads;
  for (int i = t; i < 18 * heads; i += 256) {
    const int hh = i / 18, k = i - hh * 18;
    const int which = k / 9, tap = k - which * 9;             // 0: inside (p, minus sign), 1: outside (n, plus sign)
    const int ns = which ? in_ : ip;
    if (ns == 0x7fffffff) continue;
    const int ys = ns / size, xs = ns - ys * size;
    const int sy = (int)(((long long)ys * Hs) / size), sx = (int)(((long long)xs * Ws) / size);
    const float sv = segmap[(((long long)b * seg_l + l) * Hs + sy) * Ws + sx];
    const float coef = which ? cbase * (1.0f - sv) : -cbase * sv;
    const int yy = ys + tap / 3 - 1, xx = xs + tap % 3 - 1;
    if (yy >= 0 && yy < size && xx >= 0 && xx < size)
      atomicAdd(&d_probs[(((long long)b * heads + hh) * n + yy * size + xx) * L + l], coef * gk[tap]);
  }
}

__global__ void local_loss_seg_finish_kernel(const float* __restrict__ scratch, const float* __restrict__ segm, float* __restrict__ loss,
                                             int B, int seg_l) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float ssum = 0.f, total = 0.f;
  for (int l = 0; l < seg_l; ++l) ssum += segm[(long long)b * seg_l + l];
  for (int l = 0; l < seg_l; ++l) total += scratch[(long long)b * seg_l + l];
  loss[b] += total / ssum;
}

// the eps-prediction loss of FullLoss.__call__ / StandardDiffusionLoss (loss.py:60-71,131-150) with EpsScaling / EpsWeighting
// (denoiser_scaling.py:16-22, denoiser_weighting.py): out = eps * (-sigma) + noised; loss_b = mean(w_b (out - target)^2), w = sigma^-2;
// seed of the reverse pass: d (mean_b loss_b) / d eps = -sigma_b * 2 w_b (out - target) / (B * C h w), written bf16 NHWC [B, hw, cpad]
// (channels >= 4 zero).  One workgroup per sample; eps fp32 NHWC [B, hw, ld_eps], noised / target fp32 NCHW [B, 4, hw].
__global__ void __launch_bounds__(256) diff_loss_grad_kernel(const float* __restrict__ eps, const float* __restrict__ noised,
                                                             const float* __restrict__ target, const float* __restrict__ sigma,
                                                             uint16_t* __restrict__ d_eps, float* __restrict__ loss, int B, int hw,
                                                             int ld_eps, int cpad) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float sg = sigma[b], w = 1.0f / (sg * sg);
  const float gscale = -sg * 2.0f * w / ((float)B * 4.0f * (float)hw);
  float acc = 0.f;
  for (int i = threadIdx.x; i < hw; i += 256) {
    float g4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float out = eps[((long long)b * hw + i) * ld_eps + c] * (-sg) + noised[((long long)b * 4 + c) * hw + i];
      const float r = out - target[((long long)b * 4 + c) * hw + i];
      acc += w * r * r;
      g4[c] = gscale * r;
    }
    uint16_t* dst = d_eps + ((long long)b * hw + i) * cpad;
    *reinterpret_cast<u32x2*>(dst) = u32x2{pack_bf16x2(g4[0], g4[1]), pack_bf16x2(g4[2], g4[3])};
    for (int c = 4; c < cpad; c += 2) *reinterpret_cast<uint32_t*>(dst + c) = 0u;
  }
  const float tot = block_sum_256(acc, red);
  if (threadIdx.x == 0) loss[b] = tot / (4.0f * (float)hw);
}

// torch.optim.AdamW (the reference's default optimizer, diffusion.py:49-51): decoupled weight decay, bias-corrected moments
__global__ void __launch_bounds__(256) adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long long n, float lr, float b1, float b2, float eps, float wd,
                                                    float bc1, float bc2_sqrt, float gscale) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * gscale;
  float pi = p[i] * (1.0f - lr * wd);
  const float mi = b1 * m[i] + (1.0f - b1) * gi;
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
  p[i] = pi;
}

}  // namespace

#define UDT_BWD_STREAM hipStream_t s = reinterpret_cast<hipStream_t>(stream); UdtProfScope prof(5, s)

extern "C" int udt_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, void* dq, void* dk, void* dv,
                            float* lse_ws, float* dsum_ws, int32_t batch, int32_t heads, int32_t n, int32_t ldq, int32_t ldo,
                            int32_t ldd, float scale, void* stream) {
  if (!q || !k || !v || !o || !d_o || !dq || !dk || !dv || !lse_ws || !dsum_ws) return UDT_ERR_BAD_ARG;
  if (batch <= 0 || heads <= 0 || n <= 0 || ldq % 8 != 0 || ldo % 8 != 0 || ldd % 4 != 0 || ldq < 64 || ldo < 64 || ldd < 64)
    return UDT_ERR_BAD_SHAPE;
  if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(o) |
       reinterpret_cast<uintptr_t>(d_o)) & 15) return UDT_ERR_BAD_ARG;
  if ((reinterpret_cast<uintptr_t>(dq) | reinterpret_cast<uintptr_t>(dk) | reinterpret_cast<uintptr_t>(dv)) & 7) return UDT_ERR_BAD_ARG;
  UDT_BWD_STREAM;
  AttnBwdParams p;
  p.q = static_cast<const uint16_t*>(q); p.k = static_cast<const uint16_t*>(k); p.v = static_cast<const uint16_t*>(v);
  p.o = static_cast<const uint16_t*>(o); p.d_o = static_cast<const uint16_t*>(d_o);
  p.dq = static_cast<uint16_t*>(dq); p.dk = static_cast<uint16_t*>(dk); p.dv = static_cast<uint16_t*>(dv);
  p.lse = lse_ws; p.dsum = dsum_ws;
  p.heads = heads; p.n = n; p.tiles = (n + 127) / 128;
  p.ldq = ldq; p.ldo = ldo; p.ldd = ldd;
  p.sq = (long long)n * ldq; p.so = (long long)n * ldo; p.sd = (long long)n * ldd;
  p.scale = scale; p.scale_log2e = scale * 1.4426950408889634f;
  const long long units = (long long)batch * heads * p.tiles;
  if (units > 0x7fffffffLL) return UDT_ERR_BAD_SHAPE;
  // at most one workgroup per CU (a batch-1 attend-and-excite call): a deep load ring, nothing else hides the latency; more: the
  // co-resident workgroups do (and the ring's registers would cost them their residency)
  static const int pf_env = getenv("UDT_ATTN_BWD_PF") ? atoi(getenv("UDT_ATTN_BWD_PF")) : 0;
  const bool deep = pf_env ? pf_env >= 4 : units <= 2 * 256;        // (MI355X: 256 CUs)
  const bool full = n % 128 == 0;
  const dim3 grid((unsigned)units), blk(256);
#define UDT_AB_LAUNCH(PF, FULLV)                                                                      \
  do {                                                                                                \
    hipLaunchKernelGGL((attn_bwd_kernel<false, PF, FULLV>), grid, blk, 0, s, p); /* LSE, D, dQ */     \
    UDT_CHECK_LAUNCH();                                                                               \
    hipLaunchKernelGGL((attn_bwd_kernel<true, PF, FULLV>), grid, blk, 0, s, p);  /* dK, dV */         \
  } while (0)
  if (deep) {
    if (full) UDT_AB_LAUNCH(4, true); else UDT_AB_LAUNCH(4, false);
  } else {
    if (full) UDT_AB_LAUNCH(1, true); else UDT_AB_LAUNCH(1, false);
  }
#undef UDT_AB_LAUNCH
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_xattn_bwd(const void* k, const void* v, const float* probs, const float* d_probs, const void* d_o, void* dq,
                             int32_t batch, int32_t heads, int32_t head_dim, int32_t nq, int32_t L, int32_t ldkv, int32_t ldo,
                             int32_t lddq, float scale, void* stream) {
  if (!k || !v || !probs || !dq || (!d_probs && !d_o)) return UDT_ERR_BAD_ARG;
  if (batch <= 0 || heads <= 0 || head_dim != 64 || nq <= 0 || L <= 0 || L > XB_L || lddq % 8 != 0 || (d_o && ldo % 8 != 0))
    return UDT_ERR_BAD_SHAPE;
  if ((reinterpret_cast<uintptr_t>(dq) | reinterpret_cast<uintptr_t>(d_o)) & 15) return UDT_ERR_BAD_ARG;
  UDT_BWD_STREAM;
  XattnBwdParams p;
  p.k = static_cast<const uint16_t*>(k); p.v = static_cast<const uint16_t*>(v); p.probs = probs; p.d_probs = d_probs;
  p.d_o = static_cast<const uint16_t*>(d_o); p.dq = static_cast<uint16_t*>(dq);
  p.heads = heads; p.nq = nq; p.L = L; p.ldkv = ldkv; p.ldo = ldo; p.lddq = lddq; p.scale = scale;
  hipLaunchKernelGGL(xattn_bwd_kernel, dim3((nq + 255) / 256, heads, batch), dim3(256), 0, s, p);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_local_loss_bwd(const float* probs, const float* mask, const float* seg_mask, const float* gkernel9, float* d_probs,
                                  float* loss_accum, float* scratch, int32_t n_samples, int32_t mask_batch, int32_t heads, int32_t size,
                                  int32_t L, int32_t seg_l, int32_t Hm, int32_t Wm, float weight, void* stream) {
  if (!probs || !mask || !seg_mask || !gkernel9 || !d_probs || !scratch) return UDT_ERR_BAD_ARG;
  if (n_samples <= 0 || mask_batch <= 0 || n_samples % mask_batch != 0 || heads <= 0 || size <= 0 || size > 120 || L <= 0 ||
      seg_l <= 0 || seg_l > L) return UDT_ERR_BAD_SHAPE;
  UDT_BWD_STREAM;
  const size_t smem = ((size_t)size * size + 16) * sizeof(float);
  hipLaunchKernelGGL(local_loss_token_kernel, dim3(seg_l, n_samples), dim3(256), smem, s, probs, mask, gkernel9, scratch, heads, size, L,
                     seg_l, Hm, Wm, mask_batch);
  UDT_CHECK_LAUNCH();
  hipLaunchKernelGGL(local_loss_finish_kernel, dim3(n_samples), dim3(256), 0, s, scratch, mask, seg_mask, gkernel9, d_probs, loss_accum,
                     heads, size, L, seg_l, Hm, Wm, mask_batch, weight);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_layernorm_bwd(const void* x, const void* dy, const float* gamma, const void* add, void* dx, int64_t rows, int32_t C,
                                 float eps, void* stream) {
  if (!x || !dy || !gamma || !dx) return UDT_ERR_BAD_ARG;
  if (rows <= 0 || C <= 0 || C % 8 != 0 || C > 2048) return UDT_ERR_BAD_SHAPE;
  UDT_BWD_STREAM;
  const unsigned blocks = (unsigned)((rows + 3) / 4);
  const int nch = (C / 8 + 63) / 64;
  const uint16_t* xp = static_cast<const uint16_t*>(x);
  const uint16_t* dp = static_cast<const uint16_t*>(dy);
  const uint16_t* ap = static_cast<const uint16_t*>(add);
  uint16_t* op = static_cast<uint16_t*>(dx);
  switch (nch) {
    case 1: hipLaunchKernelGGL(layernorm_bwd_kernel<1>, dim3(blocks), dim3(256), 0, s, xp, dp, gamma, ap, op, (long long)rows, C, eps); break;
    case 2: hipLaunchKernelGGL(layernorm_bwd_kernel<2>, dim3(blocks), dim3(256), 0, s, xp, dp, gamma, ap, op, (long long)rows, C, eps); break;
    case 3: hipLaunchKernelGGL(layernorm_bwd_kernel<3>, dim3(blocks), dim3(256), 0, s, xp, dp, gamma, ap, op, (long long)rows, C, eps); break;
    case 4: hipLaunchKernelGGL(layernorm_bwd_kernel<4>, dim3(blocks), dim3(256), 0, s, xp, dp, gamma, ap, op, (long long)rows, C, eps); break;
    default: return UDT_ERR_BAD_SHAPE;
  }
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_gn_bwd(const void* x, const void* dy, const float* gamma, const float* beta, const void* add, void* dx,
                          float* partials, int32_t B, int32_t HW, int32_t C, int32_t groups, float eps, int32_t silu, void* stream) {
  if (!x || !dy || !gamma || !beta || !dx) return UDT_ERR_BAD_ARG;
  if (B <= 0 || HW <= 0 || C <= 0 || groups <= 0 || C % groups != 0 || (C / groups) % 2 != 0 || (long long)HW * (C / groups) >= (1LL << 31))
    return UDT_ERR_BAD_SHAPE;
  if (partials && C % 8 == 0 && C <= 4096 && 256 % groups == 0) {
    // chunked form: partials fp32 [2][B][udt_gn_nchunks(HW, C)][groups][2]
    const int nch = udt_gn_nchunks(HW, C);
    float* part2 = partials + (size_t)B * nch * groups * 2;
    const int rc = udt_gn_stats(x, nullptr, partials, B, HW, C, 0, groups, stream);
    if (rc != UDT_OK) return rc;
    UDT_BWD_STREAM;
    const int c8 = C / 8;
    const int R = c8 <= 256 ? 256 / c8 : 1;
    const size_t smem1 = 512 * sizeof(double) + (size_t)(2 * groups + 2 * R * C) * sizeof(float);
    hipLaunchKernelGGL(gn_bwd_sums_kernel, dim3(nch, B), dim3(256), smem1, s, static_cast<const uint16_t*>(x),
                       static_cast<const uint16_t*>(dy), partials, gamma, beta, part2, (long long)HW, C, groups, nch, eps, silu);
    UDT_CHECK_LAUNCH();
    const long long total = (long long)HW * c8, chunks_per_wg = 1024;
    const int blocks = (int)((total + chunks_per_wg - 1) / chunks_per_wg);
    const size_t smem2 = 512 * sizeof(double) + (size_t)(4 * groups + 4 * C) * sizeof(float);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(blocks, B), dim3(256), smem2, s, static_cast<const uint16_t*>(x),
                       static_cast<const uint16_t*>(dy), static_cast<const uint16_t*>(add), static_cast<uint16_t*>(dx), partials, part2,
                       gamma, beta, (long long)HW, C, groups, nch, eps, silu, chunks_per_wg);
    UDT_CHECK_LAUNCH();
    return UDT_OK;
  }
  UDT_BWD_STREAM;
  hipLaunchKernelGGL(gn_bwd_kernel, dim3((unsigned)(B * groups)), dim3(256), 0, s, static_cast<const uint16_t*>(x),
                     static_cast<const uint16_t*>(dy), gamma, beta, static_cast<const uint16_t*>(add), static_cast<uint16_t*>(dx), HW, C,
                     groups, eps, silu);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_geglu_fwd(const void* ag, void* out, int64_t rows, int32_t inner, void* stream) {
  if (!ag || !out) return UDT_ERR_BAD_ARG;
  if (rows <= 0 || inner <= 0 || inner % 8 != 0) return UDT_ERR_BAD_SHAPE;
  UDT_BWD_STREAM;
  const long long n8 = rows * (inner / 8);
  hipLaunchKernelGGL(geglu_kernel<false>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, static_cast<const uint16_t*>(ag), nullptr,
                     static_cast<uint16_t*>(out), n8, inner / 8);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_geglu_bwd(const void* ag, const void* dy, void* dag, int64_t rows, int32_t inner, void* stream) {
  if (!ag || !dy || !dag) return UDT_ERR_BAD_ARG;
  if (rows <= 0 || inner <= 0 || inner % 8 != 0) return UDT_ERR_BAD_SHAPE;
  UDT_BWD_STREAM;
  const long long n8 = rows * (inner / 8);
  hipLaunchKernelGGL(geglu_kernel<true>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, static_cast<const uint16_t*>(ag),
                     static_cast<const uint16_t*>(dy), static_cast<uint16_t*>(dag), n8, inner / 8);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_sum2x2_bf16(const void* dy, void* dx, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
  if (!dy || !dx) return UDT_ERR_BAD_ARG;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 != 0) return UDT_ERR_BAD_SHAPE;
  UDT_BWD_STREAM;
  const long long n8 = (long long)B * H * W * (C / 8);
  hipLaunchKernelGGL(sum2x2_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, static_cast<const uint16_t*>(dy),
                     static_cast<uint16_t*>(dx), H, W, C / 8, n8);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_axpy_f32(float* x, const float* y, float a, int64_t n, void* stream) {
  if (!x || !y) return UDT_ERR_BAD_ARG;
  if (n <= 0) return UDT_ERR_BAD_SHAPE;
  UDT_BWD_STREAM;
  hipLaunchKernelGGL(axpy_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, a, (long long)n);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_center_tokens(const float* x, void* out, int32_t B, int32_t L, int32_t D, void* stream) {
  if (!x || !out) return UDT_ERR_BAD_ARG;
  if (B <= 0 || L <= 0 || D <= 0) return UDT_ERR_BAD_SHAPE;
  UDT_BWD_STREAM;
  hipLaunchKernelGGL(center_tokens_kernel, dim3((D + 255) / 256, B), dim3(256), 0, s, x, static_cast<uint16_t*>(out), L, D);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_transpose_bf16(const void* in, void* out, int32_t R, int32_t C, int32_t ld, int32_t Rp, void* stream) {
  if (!in || !out) return UDT_ERR_BAD_ARG;
  if (R <= 0 || C <= 0 || ld < C || Rp < R || Rp % 64 != 0) return UDT_ERR_BAD_SHAPE;
  UDT_BWD_STREAM;
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3(Rp / 64, (C + 63) / 64), dim3(256), 0, s, static_cast<const uint16_t*>(in),
                     static_cast<uint16_t*>(out), R, C, ld, Rp);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_reduce_rows_f32(const float* in, float* out, int32_t P, int64_t n, int32_t accumulate, void* stream) {
  if (!in || !out) return UDT_ERR_BAD_ARG;
  if (P <= 0 || n <= 0) return UDT_ERR_BAD_SHAPE;
  UDT_BWD_STREAM;
  hipLaunchKernelGGL(reduce_rows_f32_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, in, out, P, (long long)n, accumulate);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

/* partial buffers of the two column-reduction launches below: udt_colparts(rows) row blocks of 4 waves each */
extern "C" int32_t udt_colparts(int64_t rows) {
  int64_t wgs = (rows + 63) / 64;             // 16 rows per wave up to 16 K rows (one workgroup per CU), more beyond
  if (wgs > 256) wgs = 256;
  if (wgs < 1) wgs = 1;
  return (int32_t)(wgs * 4);
}

extern "C" int udt_colsum_bf16(const void* x, float* partials, float* out, int64_t rows, int32_t C, void* stream) {
  if (!x || !partials || !out) return UDT_ERR_BAD_ARG;
  if (rows <= 0 || C <= 0 || C % 2 != 0) return UDT_ERR_BAD_SHAPE;
  UDT_BWD_STREAM;
  const int parts = udt_colparts(rows), wgs = parts / 4;
  const int rpw = (int)((rows + wgs - 1) / wgs);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(wgs), dim3(256), 0, s, static_cast<const uint16_t*>(x), partials, (long long)rows, C, rpw);
  UDT_CHECK_LAUNCH();
  hipLaunchKernelGGL(reduce_rows_f32_kernel, dim3((C + 15) / 16), dim3(256), 0, s, partials, out, parts, (long long)C, 0);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_ln_param_grad(const void* x, const void* dy, float* partials, float* dgamma_dbeta, int64_t rows, int32_t C, float eps,
                                 void* stream) {
  if (!x || !dy || !partials || !dgamma_dbeta) return UDT_ERR_BAD_ARG;
  if (rows <= 0 || C <= 0 || C % 8 != 0 || C > 2048) return UDT_ERR_BAD_SHAPE;
  UDT_BWD_STREAM;
  const int parts = udt_colparts(rows), wgs = parts / 4;
  const int rpw = (int)((rows + wgs - 1) / wgs);
  const int nch = (C / 8 + 63) / 64;
  const uint16_t* xp = static_cast<const uint16_t*>(x);
  const uint16_t* dp = static_cast<const uint16_t*>(dy);
  switch (nch) {
    case 1: hipLaunchKernelGGL(ln_param_grad_kernel<1>, dim3(wgs), dim3(256), 0, s, xp, dp, partials, (long long)rows, C, eps, rpw); break;
    case 2: hipLaunchKernelGGL(ln_param_grad_kernel<2>, dim3(wgs), dim3(256), 0, s, xp, dp, partials, (long long)rows, C, eps, rpw); break;
    case 3: hipLaunchKernelGGL(ln_param_grad_kernel<3>, dim3(wgs), dim3(256), 0, s, xp, dp, partials, (long long)rows, C, eps, rpw); break;
    case 4: hipLaunchKernelGGL(ln_param_grad_kernel<4>, dim3(wgs), dim3(256), 0, s, xp, dp, partials, (long long)rows, C, eps, rpw); break;
    default: return UDT_ERR_BAD_SHAPE;
  }
  UDT_CHECK_LAUNCH();
  hipLaunchKernelGGL(reduce_rows_f32_kernel, dim3((2 * C + 15) / 16), dim3(256), 0, s, partials, dgamma_dbeta, parts, (long long)2 * C, 0);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int32_t udt_wgrad_splits(int64_t R, int32_t N, int32_t K) {
  if (R <= 0 || N <= 0 || K <= 0) return 0;
  const long long tiles = (long long)((N + 127) / 128) * ((K + 127) / 128);
  long long S = (512 + tiles - 1) / tiles;                      // about two workgroups per CU
  const long long smax = (R + 63) / 64;                         // at least two steps per range
  if (S > smax) S = smax;
  if (S < 1) S = 1;
  return (int32_t)S;
}

extern "C" int udt_wgrad_bf16(const void* dy, const void* x, float* dw, float* partials, int64_t R, int32_t N, int32_t K, int32_t ldy,
                              int32_t ldx, void* stream) {
  if (!dy || !x || !dw) return UDT_ERR_BAD_ARG;
  if (R <= 0 || R > 0x7fffff00LL || N <= 0 || K <= 0 || N % 8 != 0 || K % 8 != 0 || ldy < N || ldx < K || ldy % 8 != 0 || ldx % 8 != 0)
    return UDT_ERR_BAD_SHAPE;
  if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x)) & 15) return UDT_ERR_BAD_ARG;
  const int S = udt_wgrad_splits(R, N, K);
  if (S > 1 && !partials) return UDT_ERR_BAD_ARG;
  UDT_BWD_STREAM;
  int rps = (int)((R + S - 1) / S);
  rps = (rps + WG_RS - 1) / WG_RS * WG_RS;
  const int S_used = (int)((R + rps - 1) / rps);                // (rounding the range up to whole steps may need fewer of them)
  float* dst = S_used > 1 ? partials : dw;
  hipLaunchKernelGGL(wgrad_kernel, dim3((K + 127) / 128, (N + 127) / 128, S_used), dim3(256), 0, s, static_cast<const uint16_t*>(dy),
                     static_cast<const uint16_t*>(x), dst, (int)R, N, K, ldy, ldx, rps);
  UDT_CHECK_LAUNCH();
  if (S_used > 1) {
    const long long n = (long long)N * K;
    hipLaunchKernelGGL(reduce_rows_f32_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, partials, dw, S_used, n, 0);
    UDT_CHECK_LAUNCH();
  }
  return UDT_OK;
}

extern "C" int32_t udt_xattn_kv_splits(int32_t nq) { return nq <= 0 ? 0 : (nq + XKV_QT - 1) / XKV_QT; }

extern "C" int udt_xattn_bwd_kv(const void* q, const void* v, const float* probs, const float* d_probs, const void* d_o, void* dk, void* dv,
                                float* partials, int32_t batch, int32_t heads, int32_t head_dim, int32_t nq, int32_t L, int32_t ldq,
                                int32_t ldkv, int32_t ldo, int32_t lddkv, float scale, void* stream) {
  if (!q || !v || !probs || !dk || !dv || !partials || (!d_probs && !d_o)) return UDT_ERR_BAD_ARG;
  if (batch <= 0 || heads <= 0 || head_dim != 64 || nq <= 0 || L <= 0 || L > XB_L) return UDT_ERR_BAD_SHAPE;
  UDT_BWD_STREAM;
  XattnBwdParams p;
  p.k = nullptr; p.v = static_cast<const uint16_t*>(v); p.probs = probs; p.d_probs = d_probs;
  p.d_o = static_cast<const uint16_t*>(d_o); p.dq = nullptr;
  p.heads = heads; p.nq = nq; p.L = L; p.ldkv = ldkv; p.ldo = ldo; p.lddq = 0; p.scale = scale;
  const int S = udt_xattn_kv_splits(nq);
  hipLaunchKernelGGL(xattn_bwd_kv_kernel, dim3(heads, batch, S), dim3(256), 0, s, p, static_cast<const uint16_t*>(q), ldq, partials, batch);
  UDT_CHECK_LAUNCH();
  const long long rows = (long long)batch * L, tot = rows * heads * 64;
  hipLaunchKernelGGL(xattn_bwd_kv_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, partials, S, rows, heads * 64,
                     static_cast<uint16_t*>(dk), static_cast<uint16_t*>(dv), lddkv);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_local_loss_seg_bwd(const float* probs, const float* seg, const float* seg_mask, const float* gkernel9, float* d_probs,
                                      float* loss_accum, float* scratch, int32_t B, int32_t heads, int32_t size, int32_t L, int32_t seg_l,
                                      int32_t Hs, int32_t Ws, float weight, void* stream) {
  if (!probs || !seg || !seg_mask || !gkernel9 || !d_probs || !scratch) return UDT_ERR_BAD_ARG;
  if (B <= 0 || heads <= 0 || size <= 0 || size > 120 || L <= 0 || seg_l <= 0 || seg_l > L) return UDT_ERR_BAD_SHAPE;
  UDT_BWD_STREAM;
  const size_t smem = ((size_t)size * size + 32) * sizeof(float);
  hipLaunchKernelGGL(local_loss_seg_bwd_kernel, dim3(seg_l, B), dim3(256), smem, s, probs, seg, seg_mask, gkernel9, d_probs, scratch, heads,
                     size, L, seg_l, Hs, Ws, weight);
  UDT_CHECK_LAUNCH();
  if (loss_accum) {
    hipLaunchKernelGGL(local_loss_seg_finish_kernel, dim3((B + 63) / 64), dim3(64), 0, s, scratch, seg_mask, loss_accum, B, seg_l);
    UDT_CHECK_LAUNCH();
  }
  return UDT_OK;
}

extern "C" int udt_diff_loss_grad(const float* eps, const float* noised, const float* target, const float* sigma, void* d_eps, float* loss,
                                  int32_t B, int32_t hw, int32_t ld_eps, int32_t cpad, void* stream) {
  if (!eps || !noised || !target || !sigma || !d_eps || !loss) return UDT_ERR_BAD_ARG;
  if (B <= 0 || hw <= 0 || ld_eps < 4 || cpad < 4 || cpad % 4 != 0) return UDT_ERR_BAD_SHAPE;
  UDT_BWD_STREAM;
  hipLaunchKernelGGL(diff_loss_grad_kernel, dim3(B), dim3(256), 0, s, eps, noised, target, sigma, static_cast<uint16_t*>(d_eps), loss, B, hw,
                     ld_eps, cpad);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int32_t step, float grad_scale, void* stream) {
  if (!p || !g || !m || !v) return UDT_ERR_BAD_ARG;
  if (n <= 0 || step <= 0) return UDT_ERR_BAD_SHAPE;
  UDT_BWD_STREAM;
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, (long long)n, lr, beta1, beta2, eps,
                     weight_decay, bc1, sqrtf(bc2), grad_scale);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}
