// norm.hip — GroupNorm (statistics + apply with optional SiLU) and LayerNorm for NHWC bf16 activations.
//
// These are HBM-bound streaming kernels: 16-byte (8 x bf16) loads per lane, fp32 statistics.
//   GroupNorm32 / SiLU : sgm/modules/diffusionmodules/util.py:258-275, openaimodel.py:183-187,218-221
//   GroupNorm eps 1e-6 : sgm/modules/attention.py:82-85 ; sgm/modules/diffusionmodules/model.py:48-52
//   LayerNorm          : sgm/modules/attention.py:297,310-311
#include "common.h"
#include <stdio.h>
#include <stdlib.h>

namespace {

constexpr int GN_MAX_C = 4096;

// grid (nchunks, B).  Thread layout: R row-groups x c8 channel-chunks (8 channels = 16 B each).
__global__ void __launch_bounds__(256) gn_stats_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ x2,
                                                       float* __restrict__ partials, long long HW, int C1, int C2,
                                                       int G, int nchunks) {
  const int C = C1 + C2;
  extern __shared__ __attribute__((aligned(16))) float gsm[];   // [R][C] sums, then [R][C] sumsq
  const int t = threadIdx.x;
  const int c8 = C >> 3;
  const int chunk = blockIdx.x;
  const int b = blockIdx.y;
  const long long rows_per_chunk = (HW + nchunks - 1) / nchunks;
  const long long p0 = chunk * rows_per_chunk;
  long long p1 = p0 + rows_per_chunk;
  if (p1 > HW) p1 = HW;

  int R, rg, cc0, cstep;
  if (c8 <= 256) {
    R = 256 / c8;
    rg = t / c8;
    cc0 = t - rg * c8;
    cstep = c8;            // one channel chunk per thread
    if (rg >= R) cc0 = c8; // inactive
  } else {
    R = 1; rg = 0; cc0 = t; cstep = 256;
  }
  float* ssum = gsm;
  float* ssq = gsm + R * C;
  for (int cc = cc0; cc < c8; cc += cstep) {
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
    const bool second = cc * 8 >= C1;
    const int cs = second ? C2 : C1;
    const uint16_t* xb = (second ? x2 : x) + (long long)b * HW * cs + (second ? cc * 8 - C1 : cc * 8);
    auto acc8 = [&](const u32x4 v) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = bf16_lo(v[j]), bb = bf16_hi(v[j]);
        s[2 * j] += a; q[2 * j] += a * a;
        s[2 * j + 1] += bb; q[2 * j + 1] += bb * bb;
      }
    };
    long long pix = p0 + rg;
    for (; pix + 3LL * R < p1; pix += 4LL * R) {          // 4 loads in flight per lane
      const u32x4 v0 = *reinterpret_cast<const u32x4*>(xb + pix * cs);
      const u32x4 v1 = *reinterpret_cast<const u32x4*>(xb + (pix + R) * cs);
      const u32x4 v2 = *reinterpret_cast<const u32x4*>(xb + (pix + 2LL * R) * cs);
      const u32x4 v3 = *reinterpret_cast<const u32x4*>(xb + (pix + 3LL * R) * cs);
      acc8(v0); acc8(v1); acc8(v2); acc8(v3);
    }
    for (; pix < p1; pix += R) acc8(*reinterpret_cast<const u32x4*>(xb + pix * cs));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ssum[rg * C + cc * 8 + j] = s[j];
      ssq[rg * C + cc * 8 + j] = q[j];
    }
  }
  __syncthreads();
  if (t < G) {
    const int cpg = C / G;
    float a = 0.f, q = 0.f;
    for (int r = 0; r < R; ++r)
      for (int c = t * cpg; c < (t + 1) * cpg; ++c) {
        a += ssum[r * C + c];
        q += ssq[r * C + c];
      }
    float* dst = partials + (((long long)b * nchunks + chunk) * G + t) * 2;
    dst[0] = a;
    dst[1] = q;
  }
}

// grid (blocks_per_sample, B); each workgroup normalises a contiguous span of one sample
__global__ void __launch_bounds__(256) gn_apply_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ x2,
                                                       uint16_t* __restrict__ y, const float* __restrict__ partials,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, long long HW, int C1, int C2,
                                                       int G, int nchunks, float eps, int act,
                                                       long long chunks_per_wg, const float* __restrict__ scsh) {
  const int C = C1 + C2;
  extern __shared__ __attribute__((aligned(16))) float asm_[];   // [C] scale, [C] shift, [G] mean, [G] rstd
  float* sc = asm_;
  float* sh = asm_ + C;
  float* gm = asm_ + 2 * C;
  float* gr = gm + G;
  const int t = threadIdx.x;
  const int b = blockIdx.y;
  const int cpg = C / G;
  const int c8 = C >> 3;
  const long long total = HW * c8;                 // 16-byte chunks in this sample
  const long long begin = (long long)blockIdx.x * chunks_per_wg;
  long long end = begin + chunks_per_wg;
  if (end > total) end = total;
  const uint16_t* xb1 = x + (long long)b * HW * C1;
  const uint16_t* xb2 = x2 ? (x2 + (long long)b * HW * C2) : nullptr;
  uint16_t* yb = y + (long long)b * HW * C;
  auto src_of = [&](long long i, int& cc) -> const uint16_t* {
    const long long pix = i / c8;
    cc = (int)(i - pix * c8);
    return (cc * 8 < C1) ? (xb1 + pix * C1 + cc * 8) : (xb2 + pix * C2 + (cc * 8 - C1));
  };
  // the first round of loads goes out BEFORE the scale / shift table is built: a workgroup's latency is one memory round
  // trip, not table + data (the kernel is a chain of dependent round trips per workgroup; 32 KiB spans of 8 rounds ran at
  // 1.5 TB/s, profiles/r03_trace_step_gn_epilogue.txt)
  long long i = begin + t;
  u32x4 v0 = {0u, 0u, 0u, 0u}, v1 = v0, v2 = v0, v3 = v0;
  int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  const bool full = i + 768 < end;
  if (full) {
    v0 = *reinterpret_cast<const u32x4*>(src_of(i, c0));
    v1 = *reinterpret_cast<const u32x4*>(src_of(i + 256, c1));
    v2 = *reinterpret_cast<const u32x4*>(src_of(i + 512, c2));
    v3 = *reinterpret_cast<const u32x4*>(src_of(i + 768, c3));
  }
  if (scsh) {
    // the per-(sample, channel) scale / shift table of udt_gn_finalize ([B][C/64][2][64]): statistics came from the
    // producers' epilogues, nothing to reduce here
    for (int c = t; c < C; c += 256) {
      const float* src = scsh + (((long long)b * (C >> 6) + (c >> 6)) * 2) * 64 + (c & 63);
      sc[c] = src[0];
      sh[c] = src[64];
    }
  } else {
    // 256 threads = G groups x (256/G) lanes; each lane sums a strided subset of the chunk partials
    double* red = reinterpret_cast<double*>(gr + G);          // [2][256] doubles
    const int lanes = 256 / G;
    const int g = t % G, ln = t / G;
    double a = 0.0, q = 0.0;
    if (ln < lanes)
      for (int k = ln; k < nchunks; k += lanes) {
        const float* src = partials + (((long long)b * nchunks + k) * G + g) * 2;
        a += (double)src[0];
        q += (double)src[1];
      }
    red[t] = a;
    red[256 + t] = q;
    __syncthreads();
    if (t < G) {
      a = 0.0; q = 0.0;
      for (int l = 0; l < lanes; ++l) { a += red[l * G + t]; q += red[256 + l * G + t]; }
      const double n = (double)HW * (double)cpg;
      const double mean = a / n;
      double var = q / n - mean * mean;
      if (var < 0.0) var = 0.0;
      gm[t] = (float)mean;
      gr[t] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    for (int c = t; c < C; c += 256) {
      const int g = c / cpg;
      const float a = gr[g] * gamma[c];
      sc[c] = a;
      sh[c] = beta[c] - gm[g] * a;
    }
  }
  __syncthreads();
  auto norm8 = [&](const u32x4 v, int cc) {
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = cc * 8 + 2 * j;
      float a = bf16_lo(v[j]) * sc[c] + sh[c];
      float bb = bf16_hi(v[j]) * sc[c + 1] + sh[c + 1];
      if (act == 1) {
        a = silu_f(a);
        bb = silu_f(bb);
      }
      o[j] = pack_bf16x2(a, bb);
    }
    return o;
  };
  if (full) {
    *reinterpret_cast<u32x4*>(yb + i * 8) = norm8(v0, c0);
    *reinterpret_cast<u32x4*>(yb + (i + 256) * 8) = norm8(v1, c1);
    *reinterpret_cast<u32x4*>(yb + (i + 512) * 8) = norm8(v2, c2);
    *reinterpret_cast<u32x4*>(yb + (i + 768) * 8) = norm8(v3, c3);
    i += 1024;
  }
  for (; i + 768 < end; i += 1024) {                       // 4 independent 16-byte loads per lane
    v0 = *reinterpret_cast<const u32x4*>(src_of(i, c0));
    v1 = *reinterpret_cast<const u32x4*>(src_of(i + 256, c1));
    v2 = *reinterpret_cast<const u32x4*>(src_of(i + 512, c2));
    v3 = *reinterpret_cast<const u32x4*>(src_of(i + 768, c3));
    *reinterpret_cast<u32x4*>(yb + i * 8) = norm8(v0, c0);
    *reinterpret_cast<u32x4*>(yb + (i + 256) * 8) = norm8(v1, c1);
    *reinterpret_cast<u32x4*>(yb + (i + 512) * 8) = norm8(v2, c2);
    *reinterpret_cast<u32x4*>(yb + (i + 768) * 8) = norm8(v3, c3);
  }
  for (; i < end; i += 256) {
    int cc;
    const u32x4 v = *reinterpret_cast<const u32x4*>(src_of(i, cc));
    *reinterpret_cast<u32x4*>(yb + i * 8) = norm8(v, cc);
  }
}

// one wave per row, 4 rows per workgroup; NCH = ceil(C / 512) 16-byte chunks per lane
template <int NCH>
__global__ void __launch_bounds__(256) layernorm_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, long long rows, int C,
                                                        float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int c8 = C >> 3;
  const uint16_t* xr = x + row * C;
  float v[NCH][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + i * 64;
    if (ch < c8) {
      const u32x4 u = *reinterpret_cast<const u32x4*>(xr + ch * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[i][2 * j] = bf16_lo(u[j]);
        v[i][2 * j + 1] = bf16_hi(u[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[i][j];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + i * 64;
    if (ch < c8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = v[i][j] - mean;
        q += d * d;
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
  const float rstd = rsqrtf(q / (float)C + eps);
  uint16_t* yr = y + row * C;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + i * 64;
    if (ch < c8) {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + ch * 8);
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(gamma + ch * 8 + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + ch * 8);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(beta + ch * 8 + 4);
      float o[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = (v[i][j] - mean) * rstd * g0[j] + b0[j];
        o[4 + j] = (v[i][4 + j] - mean) * rstd * g1[j] + b1[j];
      }
      u32x4 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]),
                  pack_bf16x2(o[6], o[7])};
      *reinterpret_cast<u32x4*>(yr + ch * 8) = pk;
    }
  }
}


// ---- one-launch GroupNorm for strips that fit the caches -----------------------------------------------------------
// A group's statistics only need the group's own channels, so a workgroup can own a STRIP = `gw` consecutive groups of one
// sample (gw chosen so that the strip is a whole number of 16-byte chunks per pixel: 80 .. 240 bytes for the UNet's
// channel counts), read it once for the statistics and a second time — now out of L2 — to normalise.  One launch and
// one HBM read instead of gn_stats + gn_apply's two launches and two HBM reads.  grid (G / gw, B), 512 threads;
// thread = (chunk j of the strip, pixel lane); deterministic reductions (no atomics): per-lane partial sums -> LDS
// matrix -> one thread per channel -> one thread per group (fp64 combine, as gn_apply).
constexpr int GNS_THREADS = 512;

__global__ void __launch_bounds__(GNS_THREADS) gn_strip_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ x2,
                                                               uint16_t* __restrict__ y, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, long long HW, int C1, int C2, int G,
                                                               int gw, float eps, int act, const float* __restrict__ st1,
                                                               int slots1, const float* __restrict__ st2, int slots2) {
  extern __shared__ __attribute__((aligned(16))) float gss[];
  const int C = C1 + C2;
  const int cpg = C / G;
  const int cw = gw * cpg;                         // channels of the strip (multiple of 8)
  const int nch = cw >> 3;                         // 16-byte chunks per pixel
  const int pstep = GNS_THREADS / nch;             // pixel lanes
  float* psum = gss;                               // [pstep][cw]
  float* psq = gss + pstep * cw;                   // [pstep][cw]
  float* scl = psq + pstep * cw;                   // [cw] scale, then [cw] shift
  float* shf = scl + cw;
  double* csum = reinterpret_cast<double*>(shf + cw + (cw & 1));   // [cw] channel totals (fp64), [cw] squares
  double* csq = csum + cw;
  const int t = threadIdx.x;
  const int b = blockIdx.y;
  const int c_base = blockIdx.x * cw;
  const int j = t % nch, pl = t / nch;
  const bool active = pl < pstep;
  const int c = c_base + 8 * j;                    // first of this thread's 8 channels (one source: C1 % 8 == 0)
  const bool second = c >= C1;
  const int cs = second ? C2 : C1;
  const uint16_t* src = (second ? x2 : x) + (long long)b * HW * cs + (second ? c - C1 : c);
  if (st1) {
    // the producers emitted this activation's column statistics from their epilogues (udt_gemm_desc.colstats, fp32
    // [slots][C][2] per source): the statistics pass over the strip is a sum over a few slots per channel
    if (t < cw) {
      const int cc = c_base + t;
      const bool sec = cc >= C1;
      const float* sp = sec ? st2 + ((long long)b * slots2 * C2 + (cc - C1)) * 2 : st1 + ((long long)b * slots1 * C1 + cc) * 2;
      const int n = sec ? slots2 : slots1;
      const long long stride = (long long)(sec ? C2 : C1) * 2;
      double a = 0.0, qq = 0.0;
      for (int k = 0; k < n; ++k) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(sp + k * stride);
        a += (double)v[0];
        qq += (double)v[1];
      }
      csum[t] = a;
      csq[t] = qq;
    }
  }
  float s[8], q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = q[i] = 0.f;
  auto acc8 = [&](const u32x4 v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a = bf16_lo(v[i]), bb = bf16_hi(v[i]);
      s[2 * i] += a; q[2 * i] += a * a;
      s[2 * i + 1] += bb; q[2 * i + 1] += bb * bb;
    }
  };
  if (active && !st1) {
    long long p = pl;
    for (; p + 3LL * pstep < HW; p += 4LL * pstep) {           // 4 loads in flight per lane
      const u32x4 v0 = *reinterpret_cast<const u32x4*>(src + p * cs);
      const u32x4 v1 = *reinterpret_cast<const u32x4*>(src + (p + pstep) * cs);
      const u32x4 v2 = *reinterpret_cast<const u32x4*>(src + (p + 2LL * pstep) * cs);
      const u32x4 v3 = *reinterpret_cast<const u32x4*>(src + (p + 3LL * pstep) * cs);
      acc8(v0); acc8(v1); acc8(v2); acc8(v3);
    }
    for (; p < HW; p += pstep) acc8(*reinterpret_cast<const u32x4*>(src + p * cs));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      psum[pl * cw + 8 * j + i] = s[i];
      psq[pl * cw + 8 * j + i] = q[i];
    }
  }
  __syncthreads();
  if (t < cw && !st1) {
    double a = 0.0, qq = 0.0;
    for (int r = 0; r < pstep; ++r) { a += (double)psum[r * cw + t]; qq += (double)psq[r * cw + t]; }
    csum[t] = a;
    csq[t] = qq;
  }
  __syncthreads();
  if (t < cw) {
    const int g0 = (t / cpg) * cpg;                // every channel thread re-adds its group's cpg totals (<= 80 adds)
    double a = 0.0, qq = 0.0;
    for (int k = 0; k < cpg; ++k) { a += csum[g0 + k]; qq += csq[g0 + k]; }
    const double n = (double)HW * (double)cpg;
    const double mean = a / n;
    double var = qq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = rstd * gamma[c_base + t];
    scl[t] = sc;
    shf[t] = beta[c_base + t] - (float)mean * sc;
  }
  __syncthreads();
  if (!active) return;
  float sc8[8], sh8[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { sc8[i] = scl[8 * j + i]; sh8[i] = shf[8 * j + i]; }
  uint16_t* dst = y + (long long)b * HW * C + c;
  auto norm8 = [&](const u32x4 v) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a = bf16_lo(v[i]) * sc8[2 * i] + sh8[2 * i];
      float bb = bf16_hi(v[i]) * sc8[2 * i + 1] + sh8[2 * i + 1];
      if (act == 1) { a = silu_f(a); bb = silu_f(bb); }
      o[i] = pack_bf16x2(a, bb);
    }
    return o;
  };
  long long p = pl;
  for (; p + 3LL * pstep < HW; p += 4LL * pstep) {
    const u32x4 v0 = *reinterpret_cast<const u32x4*>(src + p * cs);
    const u32x4 v1 = *reinterpret_cast<const u32x4*>(src + (p + pstep) * cs);
    const u32x4 v2 = *reinterpret_cast<const u32x4*>(src + (p + 2LL * pstep) * cs);
    const u32x4 v3 = *reinterpret_cast<const u32x4*>(src + (p + 3LL * pstep) * cs);
    *reinterpret_cast<u32x4*>(dst + p * C) = norm8(v0);
    *reinterpret_cast<u32x4*>(dst + (p + pstep) * C) = norm8(v1);
    *reinterpret_cast<u32x4*>(dst + (p + 2LL * pstep) * C) = norm8(v2);
    *reinterpret_cast<u32x4*>(dst + (p + 3LL * pstep) * C) = norm8(v3);
  }
  for (; p < HW; p += pstep) *reinterpret_cast<u32x4*>(dst + p * C) = norm8(*reinterpret_cast<const u32x4*>(src + p * cs));
}

// groups per strip so that a strip is a whole number of 16-byte chunks per pixel; 0 = shape not served by the strip kernel
int gn_strip_groups(long long HW, int C1, int C2, int G) {
  const int C = C1 + C2;
  if (G <= 0 || C % G != 0 || C1 % 8 != 0 || C2 % 8 != 0) return 0;
  const int cpg = C / G;
  int gw = 1;
  while ((gw * cpg) % 8 != 0) gw *= 2;
  if (gw > 8 || G % gw != 0) return 0;
  const int cw = gw * cpg;
  if (cw > 256) return 0;
  // Measured on MI355X (B = 4, one launch stream): 64x64x320 strips (320 KiB) 41 us vs 28 us for the stats + apply pair,
  // 32x32x640 (80 KiB) 22 vs 23.5, 16x16x1280 (20 KiB) 12 vs 20.6, 8x8x1280 (5 KiB) 12 vs 20: a strip is walked by ONE
  // workgroup, so only small strips (the launch-latency-bound levels) take this kernel
  constexpr long long limit_kb = 64;
  const long long bytes = HW * cw * 2;
  if (bytes > (limit_kb << 10)) return 0;
  return gw;
}


// ---- GroupNorm statistics from producer epilogues -------------------------------------------------------------
// grid (G, B): one workgroup per (sample, group).  Thread t = (slot lane t / cpg, channel t % cpg): the slot lanes share
// the sample's slots of a channel (all loads of a workgroup in flight at once — the kernel is pure latency), the
// channel totals and the group totals are combined in fp64 (as gn_apply does), and the group's channels write the
// chunk-blocked scale / shift table the patch-staged convolution DMAs: scsh[b][c / 64][0][c % 64] = scale,
// [1][c % 64] = shift.
__global__ void __launch_bounds__(256) gn_finalize_kernel(const float* __restrict__ st1, int slots1, int C1,
                                                          const float* __restrict__ st2, int slots2, int C2,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ scsh, long long HW, int G, float eps) {
  __shared__ double ssum[256], ssq[256];
  __shared__ double csum[128], csq[128];
  __shared__ float gstat[2];
  const int C = C1 + C2;
  const int cpg = C / G;                              // <= 128
  const int b = blockIdx.y;
  const int g = blockIdx.x;
  const int t = threadIdx.x;
  const int nsl = 256 / cpg;                          // slot lanes
  const int sl = t / cpg;
  const int j = t - sl * cpg;
  const int c = g * cpg + j;
  double a = 0.0, q = 0.0;
  if (sl < nsl) {
    const bool second = c >= C1;
    const float* src = second ? st2 + ((long long)b * slots2 * C2 + (c - C1)) * 2 : st1 + ((long long)b * slots1 * C1 + c) * 2;
    const int n = second ? slots2 : slots1;
    const long long stride = (long long)(second ? C2 : C1) * 2;
    for (int k = sl; k < n; k += nsl) {
      const f32x2 v = *reinterpret_cast<const f32x2*>(src + k * stride);
      a += (double)v[0];
      q += (double)v[1];
    }
  }
  ssum[t] = a;
  ssq[t] = q;
  __syncthreads();
  if (t < cpg) {
    double ca = 0.0, cq = 0.0;
    for (int k = 0; k < nsl; ++k) { ca += ssum[k * cpg + t]; cq += ssq[k * cpg + t]; }
    csum[t] = ca;
    csq[t] = cq;
  }
  __syncthreads();
  if (t == 0) {
    double ga = 0.0, gq = 0.0;
    for (int k = 0; k < cpg; ++k) { ga += csum[k]; gq += csq[k]; }
    const double n = (double)HW * (double)cpg;
    const double mean = ga / n;
    double var = gq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    gstat[0] = (float)mean;
    gstat[1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  if (t < cpg) {
    const int cc = g * cpg + t;
    const float sc = gstat[1] * gamma[cc];
    const float sh = beta[cc] - gstat[0] * sc;
    float* dst = scsh + (((long long)b * (C >> 6) + (cc >> 6)) * 2) * 64 + (cc & 63);
    dst[0] = sc;
    dst[64] = sh;
  }
}

}  // namespace

extern "C" int udt_gn_finalize(const float* stats1, int32_t slots1, int32_t C1, const float* stats2, int32_t slots2,
                               int32_t C2, const float* gamma, const float* beta, float* scsh, int32_t B, int64_t HW,
                               int32_t G, float eps, void* stream) {
  if (!stats1 || !gamma || !beta || !scsh || (C2 > 0 && !stats2)) return UDT_ERR_BAD_ARG;
  if (B <= 0 || HW <= 0 || slots1 <= 0 || C1 <= 0 || C2 < 0 || (C2 > 0 && slots2 <= 0) || G <= 0 || G > 256)
    return UDT_ERR_BAD_SHAPE;
  const int C = C1 + C2;
  if (C % 64 != 0 || C % G != 0 || C / G > 128) return UDT_ERR_BAD_SHAPE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  UdtProfScope prof(4, s);
  if (prof.rec) {
    char tag[96];
    snprintf(tag, sizeof(tag), "gn_finalize B=%d HW=%lld C=%d+%d slots=%d", B, (long long)HW, C1, C2, slots1);
    udt_prof_tag(prof.rec, tag);
  }
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(G, B), dim3(256), 0, s, stats1, slots1, C1, stats2, slots2, C2, gamma, beta, scsh,
                     (long long)HW, G, eps);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int32_t udt_gn_strip_ok(int32_t B, int64_t HW, int32_t C1, int32_t C2, int32_t G) {
  return (B > 0 && gn_strip_groups(HW, C1, C2, G) > 0) ? 1 : 0;
}

static int gn_strip_impl(const void* x, const void* x2, void* y, const float* st1, int slots1, const float* st2, int slots2,
                         const float* gamma, const float* beta, int32_t B, int64_t HW, int32_t C1, int32_t C2, int32_t G, float eps,
                         int32_t act, void* stream) {
  if (!x || !y || !gamma || !beta || (C2 > 0 && !x2)) return UDT_ERR_BAD_ARG;
  if (B <= 0 || HW <= 0 || C1 <= 0 || C2 < 0) return UDT_ERR_BAD_SHAPE;
  if (st1 && (slots1 <= 0 || (C2 > 0 && (!st2 || slots2 <= 0)))) return UDT_ERR_BAD_ARG;
  const int gw = gn_strip_groups(HW, C1, C2, G);
  if (gw <= 0) return UDT_ERR_BAD_SHAPE;
  const int C = C1 + C2;
  const int cw = gw * (C / G);
  const int pstep = GNS_THREADS / (cw / 8);
  const size_t smem = (size_t)(2 * pstep * cw + 2 * cw + 1) * sizeof(float) + (size_t)2 * cw * sizeof(double) + 8;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  UdtProfScope prof(4, s);
  if (prof.rec) {
    char tag[96];
    snprintf(tag, sizeof(tag), "gn_strip%s B=%d HW=%lld C=%d+%d gw=%d", st1 ? "_stats" : "", B, (long long)HW, C1, C2, gw);
    udt_prof_tag(prof.rec, tag);
  }
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gn_strip_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) return udt_set_hip_error(e);
    attr_set = true;
  }
  hipLaunchKernelGGL(gn_strip_kernel, dim3(G / gw, B), dim3(GNS_THREADS), smem, s, reinterpret_cast<const uint16_t*>(x),
                     reinterpret_cast<const uint16_t*>(x2), reinterpret_cast<uint16_t*>(y), gamma, beta, (long long)HW, C1, C2, G, gw,
                     eps, act, st1, slots1, st2, slots2);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_gn_strip(const void* x, const void* x2, void* y, const float* gamma, const float* beta, int32_t B, int64_t HW,
                            int32_t C1, int32_t C2, int32_t G, float eps, int32_t act, void* stream) {
  return gn_strip_impl(x, x2, y, nullptr, 0, nullptr, 0, gamma, beta, B, HW, C1, C2, G, eps, act, stream);
}

extern "C" int udt_gn_strip_stats(const void* x, const void* x2, void* y, const float* stats1, int32_t slots1, const float* stats2,
                                  int32_t slots2, const float* gamma, const float* beta, int32_t B, int64_t HW, int32_t C1,
                                  int32_t C2, int32_t G, float eps, int32_t act, void* stream) {
  if (!stats1) return UDT_ERR_BAD_ARG;
  return gn_strip_impl(x, x2, y, stats1, slots1, stats2, slots2, gamma, beta, B, HW, C1, C2, G, eps, act, stream);
}

extern "C" int32_t udt_gn_nchunks(int64_t HW, int32_t C) {
  (void)C;
  int64_t n = HW / 16;            // >= 16 pixels per workgroup, up to 128 chunks per sample
  if (n < 1) n = 1;
  if (n > 128) n = 128;
  return (int32_t)n;
}

extern "C" int udt_gn_stats(const void* x, const void* x2, float* partials, int32_t B, int64_t HW, int32_t C1,
                            int32_t C2, int32_t G, void* stream) {
  if (!x || !partials || (C2 > 0 && !x2)) return UDT_ERR_BAD_ARG;
  if (C2 < 0 || C1 <= 0 || C1 % 8 != 0 || C2 % 8 != 0) return UDT_ERR_BAD_SHAPE;
  const int C = C1 + C2;
  if (B <= 0 || HW <= 0 || G <= 0 || G > 256 || C % G != 0 || C > GN_MAX_C) return UDT_ERR_BAD_SHAPE;
  const int nchunks = udt_gn_nchunks(HW, C);
  const int c8 = C / 8;
  const int R = c8 <= 256 ? 256 / c8 : 1;
  const size_t smem = (size_t)2 * R * C * sizeof(float);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  UdtProfScope prof(4, s);
  if (prof.rec) {
    char tag[96];
    snprintf(tag, sizeof(tag), "gn_stats B=%d HW=%lld C=%d", B, (long long)HW, C);
    udt_prof_tag(prof.rec, tag);
  }
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunks, B), dim3(256), smem, s, reinterpret_cast<const uint16_t*>(x),
                     reinterpret_cast<const uint16_t*>(x2), partials, (long long)HW, C1, C2, G, nchunks);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_gn_apply(const void* x, const void* x2, void* y, const float* partials, const float* gamma,
                            const float* beta, int32_t B, int64_t HW, int32_t C1, int32_t C2, int32_t G, float eps,
                            int32_t act, void* stream) {
  if (!x || !y || !partials || !gamma || !beta || (C2 > 0 && !x2)) return UDT_ERR_BAD_ARG;
  if (C2 < 0 || C1 <= 0 || C1 % 8 != 0 || C2 % 8 != 0) return UDT_ERR_BAD_SHAPE;
  const int C = C1 + C2;
  if (B <= 0 || HW <= 0 || G <= 0 || G > 256 || C % G != 0 || C > GN_MAX_C) return UDT_ERR_BAD_SHAPE;
  const int nchunks = udt_gn_nchunks(HW, C);
  const long long total = (long long)HW * (C / 8);
  const long long chunks_per_wg = 1024;   // 16 KiB of bf16 per workgroup: one round of four 16-byte loads per lane
  const int blocks = (int)((total + chunks_per_wg - 1) / chunks_per_wg);
  if (256 % G != 0) return UDT_ERR_BAD_SHAPE;
  const size_t smem = (size_t)(2 * C + 2 * G) * sizeof(float) + 8 + 512 * sizeof(double);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  UdtProfScope prof(4, s);
  if (prof.rec) {
    char tag[96];
    snprintf(tag, sizeof(tag), "gn_apply B=%d HW=%lld C=%d", B, (long long)HW, C);
    udt_prof_tag(prof.rec, tag);
  }
  hipLaunchKernelGGL(gn_apply_kernel, dim3(blocks, B), dim3(256), smem, s, reinterpret_cast<const uint16_t*>(x),
                     reinterpret_cast<const uint16_t*>(x2), reinterpret_cast<uint16_t*>(y), partials, gamma, beta,
                     (long long)HW, C1, C2, G, nchunks, eps, act, chunks_per_wg, static_cast<const float*>(nullptr));
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_gn_apply_scsh(const void* x, const void* x2, void* y, const float* scsh, int32_t B, int64_t HW, int32_t C1,
                                 int32_t C2, int32_t act, void* stream) {
  if (!x || !y || !scsh || (C2 > 0 && !x2)) return UDT_ERR_BAD_ARG;
  if (C2 < 0 || C1 <= 0 || C1 % 8 != 0 || C2 % 8 != 0) return UDT_ERR_BAD_SHAPE;
  const int C = C1 + C2;
  if (B <= 0 || HW <= 0 || C % 64 != 0 || C > GN_MAX_C) return UDT_ERR_BAD_SHAPE;
  const long long total = (long long)HW * (C / 8);
  const long long chunks_per_wg = 1024;   // 16 KiB of bf16 per workgroup: one round of four 16-byte loads per lane
  const int blocks = (int)((total + chunks_per_wg - 1) / chunks_per_wg);
  const size_t smem = (size_t)(2 * C) * sizeof(float);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  UdtProfScope prof(4, s);
  if (prof.rec) {
    char tag[96];
    snprintf(tag, sizeof(tag), "gn_apply_scsh B=%d HW=%lld C=%d", B, (long long)HW, C);
    udt_prof_tag(prof.rec, tag);
  }
  hipLaunchKernelGGL(gn_apply_kernel, dim3(blocks, B), dim3(256), smem, s, reinterpret_cast<const uint16_t*>(x),
                     reinterpret_cast<const uint16_t*>(x2), reinterpret_cast<uint16_t*>(y), static_cast<const float*>(nullptr),
                     static_cast<const float*>(nullptr), static_cast<const float*>(nullptr), (long long)HW, C1, C2, 1, 1, 0.f, act,
                     chunks_per_wg, scsh);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}



extern "C" int udt_layernorm(const void* x, void* y, const float* gamma, const float* beta, int64_t rows, int32_t C,
                             float eps, void* stream) {
  if (!x || !y || !gamma || !beta) return UDT_ERR_BAD_ARG;
  if (rows <= 0 || C <= 0 || C % 8 != 0 || C > 4096) return UDT_ERR_BAD_SHAPE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nch = (C + 511) / 512;
  const unsigned blocks = (unsigned)((rows + 3) / 4);
  const uint16_t* xp = reinterpret_cast<const uint16_t*>(x);
  uint16_t* yp = reinterpret_cast<uint16_t*>(y);
  UdtProfScope prof(4, s);
#define UDT_LN_CASE(N)                                                                                        \
  case N:                                                                                                     \
    hipLaunchKernelGGL(layernorm_kernel<N>, dim3(blocks), dim3(256), 0, s, xp, yp, gamma, beta, (long long)rows, \
                       C, eps);                                                                               \
    break;
  switch (nch) {
    UDT_LN_CASE(1) UDT_LN_CASE(2) UDT_LN_CASE(3) UDT_LN_CASE(4) UDT_LN_CASE(5) UDT_LN_CASE(6) UDT_LN_CASE(7)
    UDT_LN_CASE(8)
    default: return UDT_ERR_BAD_SHAPE;
  }
#undef UDT_LN_CASE
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}
