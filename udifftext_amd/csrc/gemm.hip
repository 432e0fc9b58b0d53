// gemm.hip — bf16 MFMA GEMM / implicit-GEMM convolution for gfx950.
//
//   out[M, N] = epilogue( A[M, K] · W[N, K]^T )          (fp32 accumulation)
//
// Both operands are K-contiguous, so one staging scheme serves linears, 1x1 convs and 3x3 convs:
// a K-tile of 64 elements (128 B per row) of BM activation rows and BN weight rows goes
// global -> LDS by 16-byte LDS-DMA (global_load_lds_dwordx4), double-buffered, one barrier per K-tile.
// The LDS image is XOR-swizzled at 16-byte granularity (slot ^= (row>>1)&7) so that the
// ds_read_b128 fragment reads of v_mfma_f32_32x32x16_bf16 (32 rows x one 16-byte k-slot per half-wave)
// are bank-conflict free; because LDS-DMA writes lane-linear, the swizzle is applied to the per-lane
// SOURCE address (guide rule 21).  For convolutions the A rows are gathered on the fly from one or
// two NHWC sources (channel concat), optionally through a nearest x2 upsample, with stride 1/2 and
// explicit top/left padding; out-of-image taps read a zero page.
//
// The MFMA is issued with the WEIGHT fragment as the A operand and the ACTIVATION fragment as B, so a
// lane ends up holding 4 consecutive output channels of one output row: 8-byte bf16 / 16-byte fp32
// stores into the row-major (NHWC) output.  UDT_GEMM_TRANSPOSED swaps the operands and stores
// out^T (4 consecutive rows per lane) — used to emit V^T for the attention kernel.
//
// Replaces: nn.Linear / nn.Conv2d call sites listed in include/udt_kernels.h.
#include "common.h"
#include <stdio.h>

namespace {

struct GemmParams {
  const uint16_t* a;
  const uint16_t* a2;
  const uint16_t* w;
  const uint16_t* zero;
  const float* bias;
  const uint16_t* res;
  const float* rowvec;
  void* out;
  int M, N, K;
  int lda, ldo, ldr, ldw;
  long long sA, sW, sO, sR;
  int Hin, Win, C1, C2, Hout, Wout, ksz, stride, pad_t, pad_l, ups;
  int rows_per_batch;
  int ldrv;
  int flags;
  float alpha;
  int tiles_m, tiles_n;
  int kt_per_split, n_ktiles;
  int split_mode;  // 1: write raw fp32 accumulators to slab blockIdx.y of `out`
};

constexpr int BK = 64;          // K elements per tile (128 bytes per row)
constexpr int ROW_BYTES = 128;

template <int BM, int BN, int WGM, int WGN, bool CONV, bool TRANS>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmParams p) {
  static_assert(WGM * WGN == 4, "4 waves per workgroup");
  static_assert(BM == WGM * 64 && BN == WGN * 64, "each wave owns a 64x64 output tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int A_BYTES = BM * ROW_BYTES;
  constexpr int B_BYTES = BN * ROW_BYTES;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int A_INSTR = BM / 32;   // 1-KiB LDS-DMA pieces per wave for the A tile
  constexpr int B_INSTR = BN / 32;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int hi = lane >> 5;

  // block -> (row tile, column tile); M fastest so concurrently running blocks share weight tiles
  const int bid = blockIdx.x;
  const int tile_m = bid % p.tiles_m;
  const int tile_n = bid / p.tiles_m;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;
  const int split = blockIdx.y;
  const int batch = blockIdx.z;

  const uint16_t* __restrict__ A = p.a + (long long)batch * p.sA;
  const uint16_t* __restrict__ A2 = p.a2;
  const uint16_t* __restrict__ W = p.w + (long long)batch * p.sW;

  // ---- per-lane staging state --------------------------------------------------------------------
  // piece ci (8 rows x 128 B) of a tile: lane -> row ci*8 + (lane>>3), physical 16-B slot lane&7,
  // logical k-slot = physical ^ ((row>>1)&7).
  const int l3 = lane >> 3;
  const int pslot = lane & 7;

  // A rows
  int a_koff[A_INSTR];                 // logical k-slot * 8 elements
  long long a_rowoff[A_INSTR];         // plain: row * lda (or -1 when out of range)
  int a_iy0[A_INSTR], a_ix0[A_INSTR], a_pixb[A_INSTR];   // conv
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int ci = wave * A_INSTR + i;
    const int row = ci * 8 + l3;
    a_koff[i] = (pslot ^ ((row >> 1) & 7)) * 8;
    const int m = m0 + row;
    if constexpr (CONV) {
      const int hw = p.Hout * p.Wout;
      const int b = m / hw;
      const int rem = m - b * hw;
      const int oy = rem / p.Wout;
      const int ox = rem - oy * p.Wout;
      a_pixb[i] = b * p.Hin * p.Win;
      a_iy0[i] = (m < p.M) ? (oy * p.stride - p.pad_t) : -100000;
      a_ix0[i] = ox * p.stride - p.pad_l;
      a_rowoff[i] = 0;
    } else {
      a_rowoff[i] = (m < p.M) ? (long long)m * p.lda : -1;
      a_iy0[i] = a_ix0[i] = a_pixb[i] = 0;
    }
  }
  // W rows
  long long w_rowoff[B_INSTR];
  int w_koff[B_INSTR];
#pragma unroll
  for (int i = 0; i < B_INSTR; ++i) {
    const int ci = wave * B_INSTR + i;
    const int row = ci * 8 + l3;
    w_koff[i] = (pslot ^ ((row >> 1) & 7)) * 8;
    const int n = n0 + row;
    w_rowoff[i] = (n < p.N) ? (long long)n * p.ldw : -1;
  }

  const int Ctot = p.C1 + p.C2;
  const int Hv = p.Hin << p.ups;
  const int Wv = p.Win << p.ups;

  auto stage = [&](int buf, int kt) {
    char* abuf = smem + buf * STAGE_BYTES;
    char* bbuf = abuf + A_BYTES;
    const int k0 = kt * BK;
    if constexpr (CONV) {
      const int tap = k0 / Ctot;
      const int c0 = k0 - tap * Ctot;
      const int ky = tap / p.ksz;
      const int kx = tap - ky * p.ksz;
      const bool second = c0 >= p.C1;
      const uint16_t* src = second ? A2 : A;
      const int cs = second ? p.C2 : p.C1;
      const int cc = second ? (c0 - p.C1) : c0;
#pragma unroll
      for (int i = 0; i < A_INSTR; ++i) {
        const int iy = a_iy0[i] + ky;
        const int ix = a_ix0[i] + kx;
        const bool ok = ((unsigned)iy < (unsigned)Hv) && ((unsigned)ix < (unsigned)Wv);
        const long long pix = (long long)a_pixb[i] + (long long)(iy >> p.ups) * p.Win + (ix >> p.ups);
        const uint16_t* g = ok ? (src + pix * cs + cc + a_koff[i]) : p.zero;
        glds16(g, abuf + (wave * A_INSTR + i) * 1024);
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_INSTR; ++i) {
        const uint16_t* g = (a_rowoff[i] >= 0) ? (A + a_rowoff[i] + k0 + a_koff[i]) : p.zero;
        glds16(g, abuf + (wave * A_INSTR + i) * 1024);
      }
    }
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
      const uint16_t* g = (w_rowoff[i] >= 0) ? (W + w_rowoff[i] + k0 + w_koff[i]) : p.zero;
      glds16(g, bbuf + (wave * B_INSTR + i) * 1024);
    }
  };

  // ---- main loop -----------------------------------------------------------------------------------
  const int wm = (WGN == 1) ? wave : (wave >> 1);
  const int wn = (WGN == 1) ? 0 : ((WGM == 1) ? wave : (wave & 1));
  const int swz = (l31 >> 1) & 7;
  const int a_frag_row = (wm * 64 + l31) * ROW_BYTES;
  const int b_frag_row = (wn * 64 + l31) * ROW_BYTES;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int kt_begin = split * p.kt_per_split;
  int kt_end = kt_begin + p.kt_per_split;
  if (kt_end > p.n_ktiles) kt_end = p.n_ktiles;

  if (kt_begin < kt_end) stage(0, kt_begin);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int cur = (kt - kt_begin) & 1;
    wait_vmcnt0();
    __syncthreads();
    if (kt + 1 < kt_end) stage(cur ^ 1, kt + 1);
    const char* abuf = smem + cur * STAGE_BYTES;
    const char* bbuf = abuf + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int slot = ((ks * 2 + hi) ^ swz) << 4;
      bf16x8_t xf[2], wf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        xf[t] = lds_read_frag(abuf + a_frag_row + t * 32 * ROW_BYTES + slot);
        wf[t] = lds_read_frag(bbuf + b_frag_row + t * 32 * ROW_BYTES + slot);
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          if constexpr (TRANS)
            acc[tm][tn] = mfma32(xf[tm], wf[tn], acc[tm][tn]);
          else
            acc[tm][tn] = mfma32(wf[tn], xf[tm], acc[tm][tn]);
        }
    }
  }

  // ---- epilogue ----------------------------------------------------------------------------------------
  const int flags = p.flags;
  if (p.split_mode) {
    // raw fp32 partial sums -> slab [split][M][N]
    float* slab = reinterpret_cast<float*>(p.out) + (long long)split * p.M * p.N;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int m = m0 + wm * 64 + tm * 32 + l31;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn * 64 + tn * 32 + q * 8 + hi * 4;
          if (m < p.M && n < p.N) {
            f32x4 v = {acc[tm][tn][q * 4 + 0], acc[tm][tn][q * 4 + 1], acc[tm][tn][q * 4 + 2],
                       acc[tm][tn][q * 4 + 3]};
            *reinterpret_cast<f32x4*>(slab + (long long)m * p.N + n) = v;
          }
        }
    }
    return;
  }

  if constexpr (TRANS) {
    // D[row = activation row][col = output channel]: lane = channel, 4 consecutive rows per quad
    uint16_t* outT = reinterpret_cast<uint16_t*>(p.out);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int n = n0 + wn * 64 + tn * 32 + l31;
        const float bias = (p.bias != nullptr && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = m0 + wm * 64 + tm * 32 + q * 8 + hi * 4;
          if (m < p.M && n < p.N) {
            const int b = m / p.rows_per_batch;
            const int tok = m - b * p.rows_per_batch;
            float v0 = acc[tm][tn][q * 4 + 0] * p.alpha + bias;
            float v1 = acc[tm][tn][q * 4 + 1] * p.alpha + bias;
            float v2 = acc[tm][tn][q * 4 + 2] * p.alpha + bias;
            float v3 = acc[tm][tn][q * 4 + 3] * p.alpha + bias;
            u32x2 pk = {pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
            long long off = ((long long)b * p.N + n) * p.rows_per_batch + tok;
            *reinterpret_cast<u32x2*>(outT + off) = pk;
          }
        }
      }
    return;
  } else {
    const uint16_t* __restrict__ R = p.res ? (p.res + (long long)batch * p.sR) : nullptr;
    if (flags & UDT_GEMM_GEGLU) {
      // wave columns [0,32) hold x, [32,64) the matching gate; output width N/2
      uint16_t* out = reinterpret_cast<uint16_t*>(p.out) + (long long)batch * p.sO;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        const int m = m0 + wm * 64 + tm * 32 + l31;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int nx = n0 + wn * 64 + q * 8 + hi * 4;        // packed index of x
          const int no = ((n0 + wn * 64) >> 1) + q * 8 + hi * 4;  // output column
          if (m < p.M && nx < p.N) {
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float x = acc[tm][0][q * 4 + r] * p.alpha;
              float g = acc[tm][1][q * 4 + r] * p.alpha;
              if (p.bias) {
                x += p.bias[nx + r];
                g += p.bias[nx + 32 + r];
              }
              o[r] = x * gelu_erf_f(g);
            }
            u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            *reinterpret_cast<u32x2*>(out + (long long)m * p.ldo + no) = pk;
          }
        }
      }
      return;
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int m = m0 + wm * 64 + tm * 32 + l31;
      const int b = (p.rowvec != nullptr) ? (m / p.rows_per_batch) : 0;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn * 64 + tn * 32 + q * 8 + hi * 4;
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

// split-K: sum the fp32 slabs and apply the epilogue (bias, rowvec, residual, activation, cast)
__global__ void __launch_bounds__(256) splitk_finalize_kernel(const float* __restrict__ slabs, int nsplit,
                                                              const GemmParams p) {
  const long long idx4 = (long long)blockIdx.x * 256 + threadIdx.x;   // one thread per 4 columns
  const int n4 = p.N >> 2;
  const long long total = (long long)p.M * n4;
  if (idx4 >= total) return;
  const int m = (int)(idx4 / n4);
  const int n = (int)(idx4 - (long long)m * n4) * 4;
  const long long mn = (long long)p.M * p.N;
  f32x4 s = *reinterpret_cast<const f32x4*>(slabs + (long long)m * p.N + n);
  for (int k = 1; k < nsplit; ++k) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(slabs + k * mn + (long long)m * p.N + n);
    s += t;
  }
  float v[4] = {s[0] * p.alpha, s[1] * p.alpha, s[2] * p.alpha, s[3] * p.alpha};
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
    const u32x2 rr = *reinterpret_cast<const u32x2*>(p.res + (long long)m * p.ldr + n);
    v[0] += bf16_lo(rr[0]);
    v[1] += bf16_hi(rr[0]);
    v[2] += bf16_lo(rr[1]);
    v[3] += bf16_hi(rr[1]);
  }
  if (p.flags & UDT_GEMM_RELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
  }
  if (p.flags & UDT_GEMM_SILU_OUT) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
  }
  if (p.flags & UDT_GEMM_OUT_F32) {
    f32x4 ov = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (long long)m * p.ldo + n) = ov;
  } else {
    u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(p.out) + (long long)m * p.ldo + n) = pk;
  }
}

struct TilePlan {
  int bm, bn;     // tile
  int splits;     // split-K factor (1 = none)
  int kt_per_split;
};

// Tile / split-K heuristic.  256 CUs; aim for >= ~2 workgroups per CU when the problem allows it.
TilePlan plan_tiles(const udt_gemm_desc* d) {
  TilePlan t;
  const int n_eff = d->N;
  // narrow outputs (N <= 64, or N a multiple of 64 but not 128 with a tall M) -> 256x64 tiles
  const bool tall = d->M >= 1024;
  if (n_eff <= 64 || (tall && (n_eff % 128) != 0 && (n_eff % 128) <= 64)) {
    t.bm = 256; t.bn = 64;
  } else {
    t.bm = 128; t.bn = 128;
  }
  if (d->flags & (UDT_GEMM_TRANSPOSED | UDT_GEMM_GEGLU)) { t.bm = 128; t.bn = 128; }
  const int tiles = ((d->M + t.bm - 1) / t.bm) * ((d->N + t.bn - 1) / t.bn) * (d->batch > 0 ? d->batch : 1);
  const int nkt = d->K / BK;
  t.splits = 1;
  const bool can_split = !(d->flags & (UDT_GEMM_TRANSPOSED | UDT_GEMM_GEGLU)) && d->batch <= 1 && (d->N % 4 == 0);
  if (can_split && tiles < 256 && nkt >= 16) {
    int want = (512 + tiles - 1) / tiles;          // reach ~512 workgroups
    int max_by_k = nkt / 8;                         // keep >= 8 K-tiles per split
    int s = want < max_by_k ? want : max_by_k;
    if (s > 16) s = 16;
    if (s >= 2) t.splits = s;
  }
  t.kt_per_split = (nkt + t.splits - 1) / t.splits;
  t.splits = (nkt + t.kt_per_split - 1) / t.kt_per_split;
  return t;
}

template <int BM, int BN, int WGM, int WGN, bool CONV, bool TRANS>
hipError_t launch_cfg(const GemmParams& p, dim3 grid, hipStream_t s) {
  constexpr int smem = 2 * (BM + BN) * ROW_BYTES;
  static bool attr_set = false;
  auto kern = gemm_kernel<BM, BN, WGM, WGN, CONV, TRANS>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, p);
  return hipGetLastError();
}

}  // namespace

extern "C" size_t udt_gemm_workspace_bytes(const udt_gemm_desc* d) {
  if (!d) return 0;
  TilePlan t = plan_tiles(d);
  if (t.splits <= 1) return 0;
  return (size_t)t.splits * (size_t)d->M * (size_t)d->N * sizeof(float);
}

extern "C" int udt_gemm(const udt_gemm_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
  if (!d || !d->a || !d->w || !d->out) return UDT_ERR_BAD_ARG;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return UDT_ERR_BAD_SHAPE;
  if (d->K % BK != 0) return UDT_ERR_BAD_SHAPE;
  if (d->N % 4 != 0) return UDT_ERR_BAD_SHAPE;
  const bool conv = (d->flags & UDT_GEMM_CONV) != 0;
  const bool trans = (d->flags & UDT_GEMM_TRANSPOSED) != 0;
  const int batch = d->batch > 0 ? d->batch : 1;
  if (conv) {
    if (d->ksize != 1 && d->ksize != 3) return UDT_ERR_BAD_SHAPE;
    if (d->C1 <= 0 || d->C1 % BK != 0 || d->C2 % BK != 0) return UDT_ERR_BAD_SHAPE;
    if (d->C2 > 0 && !d->a2) return UDT_ERR_BAD_ARG;
    if (d->K != d->ksize * d->ksize * (d->C1 + d->C2)) return UDT_ERR_BAD_SHAPE;
    if (d->Hout <= 0 || d->Wout <= 0 || d->M % (d->Hout * d->Wout) != 0) return UDT_ERR_BAD_SHAPE;
    if (d->stride != 1 && d->stride != 2) return UDT_ERR_BAD_SHAPE;
    if (trans || batch != 1) return UDT_ERR_BAD_ARG;
  } else {
    if (d->lda < d->K || d->lda % 8 != 0) return UDT_ERR_BAD_SHAPE;
  }
  if (trans) {
    if (d->rows_per_batch <= 0 || d->rows_per_batch % 4 != 0 || d->M % d->rows_per_batch != 0)
      return UDT_ERR_BAD_SHAPE;
    if (d->flags & (UDT_GEMM_OUT_F32 | UDT_GEMM_GEGLU | UDT_GEMM_RELU | UDT_GEMM_SILU_OUT)) return UDT_ERR_BAD_ARG;
    if (d->residual || d->rowvec || batch != 1) return UDT_ERR_BAD_ARG;
  } else {
    if (d->ldo % 4 != 0) return UDT_ERR_BAD_SHAPE;
    if (d->residual && d->ldr % 4 != 0) return UDT_ERR_BAD_SHAPE;
  }
  if (d->flags & UDT_GEMM_GEGLU) {
    if (d->N % 64 != 0 || (d->flags & UDT_GEMM_OUT_F32) || d->residual || d->rowvec) return UDT_ERR_BAD_ARG;
  }
  if (d->rowvec && d->rows_per_batch <= 0) return UDT_ERR_BAD_ARG;

  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  TilePlan t = plan_tiles(d);

  GemmParams p;
  p.a = reinterpret_cast<const uint16_t*>(d->a);
  p.a2 = reinterpret_cast<const uint16_t*>(d->a2);
  p.w = reinterpret_cast<const uint16_t*>(d->w);
  p.zero = udt_zero_page();
  if (!p.zero) return UDT_ERR_HIP;
  p.bias = d->bias;
  p.res = reinterpret_cast<const uint16_t*>(d->residual);
  p.rowvec = d->rowvec;
  p.out = d->out;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.lda = d->lda; p.ldo = d->ldo; p.ldr = d->ldr;
  p.ldw = d->ldw > 0 ? d->ldw : d->K;
  if (p.ldw < d->K || p.ldw % 8 != 0) return UDT_ERR_BAD_SHAPE;
  p.sA = d->stride_a; p.sW = d->stride_w; p.sO = d->stride_out; p.sR = d->stride_res;
  p.Hin = d->Hin; p.Win = d->Win; p.C1 = d->C1; p.C2 = d->C2; p.Hout = d->Hout; p.Wout = d->Wout;
  p.ksz = d->ksize; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l; p.ups = d->upsample ? 1 : 0;
  p.rows_per_batch = d->rows_per_batch > 0 ? d->rows_per_batch : d->M;
  p.ldrv = d->ld_rowvec > 0 ? d->ld_rowvec : d->N;
  p.flags = d->flags;
  p.alpha = d->alpha;
  p.tiles_m = (d->M + t.bm - 1) / t.bm;
  p.tiles_n = (d->N + t.bn - 1) / t.bn;
  p.n_ktiles = d->K / BK;
  p.kt_per_split = t.kt_per_split;
  p.split_mode = t.splits > 1 ? 1 : 0;

  if (p.split_mode) {
    const size_t need = (size_t)t.splits * (size_t)d->M * (size_t)d->N * sizeof(float);
    if (!workspace || workspace_bytes < need) return UDT_ERR_WORKSPACE;
    p.out = workspace;
  }

  const int cls = (conv && d->ksize == 3) ? 0 : 1;
  UdtProfScope prof(cls, s);
  if (prof.rec) {
    char tag[96];
    snprintf(tag, sizeof(tag), "gemm M=%d N=%d K=%d conv=%d ks=%d fl=0x%x tile=%dx%d split=%d b=%d", d->M, d->N, d->K,
             conv ? 1 : 0, d->ksize, d->flags, t.bm, t.bn, t.splits, batch);
    udt_prof_tag(prof.rec, tag);
  }

  dim3 grid(p.tiles_m * p.tiles_n, t.splits, batch);
  hipError_t e;
  if (t.bm == 256) {
    e = conv ? launch_cfg<256, 64, 4, 1, true, false>(p, grid, s) : launch_cfg<256, 64, 4, 1, false, false>(p, grid, s);
  } else if (trans) {
    e = launch_cfg<128, 128, 2, 2, false, true>(p, grid, s);
  } else {
    e = conv ? launch_cfg<128, 128, 2, 2, true, false>(p, grid, s) : launch_cfg<128, 128, 2, 2, false, false>(p, grid, s);
  }
  if (e != hipSuccess) return udt_set_hip_error(e);

  if (p.split_mode) {
    GemmParams pf = p;
    pf.out = d->out;
    const long long total4 = (long long)d->M * (d->N / 4);
    const int blocks = (int)((total4 + 255) / 256);
    hipLaunchKernelGGL(splitk_finalize_kernel, dim3(blocks), dim3(256), 0, s,
                       reinterpret_cast<const float*>(workspace), t.splits, pf);
    UDT_CHECK_LAUNCH();
  }
  return UDT_OK;
}
