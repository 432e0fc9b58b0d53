// gemm.hip — bf16 MFMA GEMM / implicit-GEMM convolution for gfx950, persistent "stream-K" scheduling.
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
// Scheduling.  The UNet's problems are mid-sized (40 .. 1300 output tiles on a 256-CU chip), so a
// one-workgroup-per-tile launch wastes 20-40 % in partial waves.  Instead the (tile, K-tile) iteration space is
// cut into G equal contiguous ranges, one per persistent workgroup (G = 2 per CU): a workgroup walks its range,
// finishing whole tiles with the fused epilogue and parking the accumulators of the (at most two) tiles it only
// partially covers in fp32 slabs; a small second kernel adds the slabs of each shared tile and runs the same epilogue.
// That is split-K where it is needed (few tiles, deep K) and plain data-parallel where it is not, with one rule.
//
// The MFMA is issued with the WEIGHT fragment as the A operand and the ACTIVATION fragment as B, so a
// lane ends up holding 4 consecutive output channels of one output row: 8-byte bf16 / 16-byte fp32
// stores into the row-major (NHWC) output.  UDT_GEMM_TRANSPOSED swaps the operands and stores
// out^T (4 consecutive rows per lane) — used to emit V^T for the attention kernel.
//
// Replaces: nn.Linear / nn.Conv2d call sites listed in include/udt_kernels.h.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <cmath>

// lean.hip: launchers of the lean / wide kernel families (argument blocks: lean_params.h, passed as opaque pointers because
// each translation unit keeps its own internal-linkage copy of the types)
hipError_t udt_lean_launch_gemm(int cfg, const void* lparams, int smem, int G, int geglu, int ln, hipStream_t s);
hipError_t udt_lean_launch_conv3(const void* c3params, hipStream_t s);

namespace {
#include "tile_common.h"

struct GemmParams {
  const uint16_t* a;
  const uint16_t* a2;
  const uint16_t* w;
  const uint16_t* zero;
  const float* bias;
  const uint16_t* res;
  const float* rowvec;
  void* out;
  const float* colscale; // fp8: per-output-channel weight scale (multiplies the accumulators), or nullptr
  const float* in_scsh;  // patch-staged convolution: GroupNorm scale/shift table of the input (conv3p.h), or nullptr
  float* colstats;       // per-(row slot, column) partial sums of the output, or nullptr
  int in_act;
  float* slabs;          // [G][2][16][256] float4 accumulator parking space
  int M, N, K;
  int lda, ldo, ldr, ldw;
  long long sA, sW, sO, sR;
  int Hin, Win, C1, C2, Hout, Wout, ksz, stride, pad_t, pad_l, ups;
  int rows_per_batch;
  int ldrv;
  int flags;
  float alpha;
  int tiles_m, tiles_n, tiles_per_batch;
  int n_block;           // N-tiles per block of the tile order (decode_tile)
  int n_ktiles;
  long long total_iters;
  int iters_per_wg;
  int G;
};

constexpr int SLAB_FLOATS = 16 * 256 * 4;   // one parked 4-wave accumulator set

// ------------------------------------------------------------------------------------------------ epilogue
// acc[tm][tn] is the wave's 64x64 block of the tile at (m0, n0); see the header comment for the layout.
template <bool TRANS>
UDT_DEVINL void epilogue(const GemmParams& p, f32x16 (&acc)[2][2], int m0, int n0, int batch, int wm, int wn,
                         int lane) {
  const int l31 = lane & 31;
  const int hi = lane >> 5;
  const int flags = p.flags;
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
          const int nx = n0 + wn * 64 + q * 8 + hi * 4;           // packed index of x
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

// tile id -> (batch, m0, n0); M fastest so that neighbouring workgroups share weight tiles
// Blocked order: the N-tiles are taken in blocks of p.n_block; inside a block the tile index runs N-fastest within an
// M-tile, then over the M-tiles.  A workgroup's consecutive tiles (and its XCD neighbours') then re-use one activation
// tile for n_block weight tiles out of L2, while the block's weight tiles (<= ~2 MiB) stay L2-resident across the
// M-tiles.  n_block = 1 is the plain M-fastest order (deep K: the weight tile alone fills the L2).
template <int BM, int BN>
UDT_DEVINL void decode_tile(const GemmParams& p, int tile, int& batch, int& m0, int& n0) {
  batch = tile / p.tiles_per_batch;
  const int t = tile - batch * p.tiles_per_batch;
  const int per_block = p.tiles_m * p.n_block;
  const int blk = t / per_block;
  const int r = t - blk * per_block;
  int nbw = p.tiles_n - blk * p.n_block;
  if (nbw > p.n_block) nbw = p.n_block;
  const int tm = r / nbw;
  const int tn = blk * p.n_block + (r - tm * nbw);
  m0 = tm * BM;
  n0 = tn * BN;
}

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
  const int l3 = lane >> 3;
  const int pslot = lane & 7;
  const int wm = (WGN == 1) ? wave : (wave >> 1);
  const int wn = (WGN == 1) ? 0 : ((WGM == 1) ? wave : (wave & 1));
  const int swz = (l31 >> 1) & 7;
  const int a_frag_row = (wm * 64 + l31) * ROW_BYTES;
  const int b_frag_row = (wn * 64 + l31) * ROW_BYTES;

  const int g = range_index(blockIdx.x, p.G);
  long long it = (long long)g * p.iters_per_wg;
  long long it_end = it + p.iters_per_wg;
  if (it_end > p.total_iters) it_end = p.total_iters;
  if (it >= it_end) return;
  const long long it_first = it;

  const int Ctot = p.C1 + p.C2;
  const int Hv = p.Hin << p.ups;
  const int Wv = p.Win << p.ups;

  // ---- per-lane staging state of the CURRENT segment -------------------------------------------------
  // piece ci (8 rows x 128 B) of a tile: lane -> row ci*8 + (lane>>3), physical 16-B slot lane&7,
  // logical k-slot = physical ^ ((row>>1)&7).
  const uint16_t* A = nullptr;
  const uint16_t* W = nullptr;
  int a_koff[A_INSTR];
  long long a_rowoff[A_INSTR];
  int a_iy0[A_INSTR], a_ix0[A_INSTR], a_pixb[A_INSTR];
  long long w_rowoff[B_INSTR];
  int w_koff[B_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int row = (wave * A_INSTR + i) * 8 + l3;
    a_koff[i] = (pslot ^ ((row >> 1) & 7)) * 8;
  }
#pragma unroll
  for (int i = 0; i < B_INSTR; ++i) {
    const int row = (wave * B_INSTR + i) * 8 + l3;
    w_koff[i] = (pslot ^ ((row >> 1) & 7)) * 8;
  }

  auto prepare = [&](int batch, int m0, int n0) {
    A = p.a + (long long)batch * p.sA;
    W = p.w + (long long)batch * p.sW;
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
      const int row = (wave * A_INSTR + i) * 8 + l3;
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
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
      const int row = (wave * B_INSTR + i) * 8 + l3;
      const int n = n0 + row;
      w_rowoff[i] = (n < p.N) ? (long long)n * p.ldw : -1;
    }
  };

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
      const uint16_t* src = second ? p.a2 : A;
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
      for (int i = 0; i < A_INSTR; ++i) {
        const uint16_t* gp = (a_rowoff[i] >= 0) ? (A + a_rowoff[i] + k0 + a_koff[i]) : p.zero;
        glds16(gp, abuf + (wave * A_INSTR + i) * 1024);
      }
    }
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
      const uint16_t* gp = (w_rowoff[i] >= 0) ? (W + w_rowoff[i] + k0 + w_koff[i]) : p.zero;
      glds16(gp, bbuf + (wave * B_INSTR + i) * 1024);
    }
  };

  // ---- walk the iteration range ---------------------------------------------------------------------------
  int tile = (int)(it / p.n_ktiles);
  int kt0 = (int)(it - (long long)tile * p.n_ktiles);
  int batch, m0, n0;
  decode_tile<BM, BN>(p, tile, batch, m0, n0);
  prepare(batch, m0, n0);
  stage(0, kt0);

  while (true) {
    int kt1 = p.n_ktiles;
    if ((long long)(kt1 - kt0) > it_end - it) kt1 = kt0 + (int)(it_end - it);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // K loop of this segment; the first tile's loads are already in flight (buffer 0)
    for (int kt = kt0; kt < kt1; ++kt) {
      const int cur = (kt - kt0) & 1;
      wait_vmcnt0();
      __syncthreads();
      if (kt + 1 < kt1) stage(cur ^ 1, kt + 1);
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

    const bool full = (kt0 == 0) && (kt1 == p.n_ktiles);
    const bool head = (it == it_first);          // segment starts at the beginning of this workgroup's range
    const int cur_batch = batch, cur_m0 = m0, cur_n0 = n0;
    it += kt1 - kt0;
    const bool more = it < it_end;
    if (more) {
      // start the next segment's first K-tile before running this segment's epilogue
      __syncthreads();                           // every wave is done reading both LDS buffers
      tile = (int)(it / p.n_ktiles);
      kt0 = (int)(it - (long long)tile * p.n_ktiles);
      decode_tile<BM, BN>(p, tile, batch, m0, n0);
      prepare(batch, m0, n0);
      stage(0, kt0);
    }
    if (full) {
      epilogue<TRANS>(p, acc, cur_m0, cur_n0, cur_batch, wm, wn, lane);
    } else {
      // park the partial accumulators: slab[g][head ? 0 : 1][i][tid] (float4), coalesced 4 KiB per i
      f32x4* slab = reinterpret_cast<f32x4*>(p.slabs) + ((long long)g * 2 + (head ? 0 : 1)) * (16 * 256);
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v = {acc[tm][tn][q * 4 + 0], acc[tm][tn][q * 4 + 1], acc[tm][tn][q * 4 + 2],
                       acc[tm][tn][q * 4 + 3]};
            slab[(tm * 8 + tn * 4 + q) * 256 + tid] = v;
          }
    }
    if (!more) break;
  }
}

// second pass of the stream-K schedule: one workgroup per output tile; tiles that were covered by a single
// workgroup were already finished by it and exit here immediately
template <int BM, int BN, int WGM, int WGN, bool TRANS>
__global__ void __launch_bounds__(256) gemm_fixup_kernel(const GemmParams p) {
  const int tile = blockIdx.x;
  const long long it0 = (long long)tile * p.n_ktiles;
  const long long it1 = it0 + p.n_ktiles;
  const int g_first = (int)(it0 / p.iters_per_wg);
  const int g_last = (int)((it1 - 1) / p.iters_per_wg);
  if (g_first == g_last) return;
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int wm = (WGN == 1) ? wave : (wave >> 1);
  const int wn = (WGN == 1) ? 0 : ((WGM == 1) ? wave : (wave & 1));
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int g = g_first; g <= g_last; ++g) {
    const long long gb = (long long)g * p.iters_per_wg;
    const int slot = (it0 <= gb) ? 0 : 1;        // segment starts at the workgroup's range start -> slot 0
    const f32x4* slab = reinterpret_cast<const f32x4*>(p.slabs) + ((long long)g * 2 + slot) * (16 * 256);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = slab[(tm * 8 + tn * 4 + q) * 256 + tid];
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[tm][tn][q * 4 + r] += v[r];
        }
  }
  int batch, m0, n0;
  decode_tile<BM, BN>(p, tile, batch, m0, n0);
  epilogue<TRANS>(p, acc, m0, n0, batch, wm, wn, lane);
}

#include "gemm8.h"
#include "conv3p.h"
#include "conv_n4.h"
#include "lean_params.h"

struct TilePlan {
  int bm, bn;
  int tiles_m, tiles_n, tiles;
  int nkt;
  long long total;
  int G, ipw;
  bool fixup;
};

// Resident workgroup slots of the CURRENT device (2 four-wave workgroups or 1 eight-wave workgroup per CU).  The
// cooperative (stream-K) kernels spin on partner workgroups, so ALL their workgroups must be resident: a launch that
// shares the device with other launch streams plans for 1/cu_share of the CUs (udt_gemm_desc.cu_share — a field of the
// call, not process state, so concurrent callers with different shares do not interfere).
constexpr int MAX_DEVICES = 16;
std::atomic<int> g_dev_cus[MAX_DEVICES];

int device_cus() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return 256;
  int v = g_dev_cus[dev].load(std::memory_order_relaxed);
  if (v == 0) {
    v = 256;
    int q = 0;
    if (hipDeviceGetAttribute(&q, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && q > 0) v = q;
    g_dev_cus[dev].store(v, std::memory_order_relaxed);
  }
  return v;
}

int resident_slots_all() { return 2 * device_cus(); }

int resident_slots(const udt_gemm_desc* d) {
  const int share = d->cu_share > 0 ? d->cu_share : 1;
  const int s = resident_slots_all() / share;
  return s < 2 ? 2 : s;
}

TilePlan plan_tiles(const udt_gemm_desc* d) {
  TilePlan t;
  const int batch = d->batch > 0 ? d->batch : 1;
  // narrow outputs (N <= 64, or a 64-wide remainder column with a tall M) -> 256x64 tiles
  const bool tall = d->M >= 1024;
  if (d->N <= 64 || (tall && (d->N % 128) != 0 && (d->N % 128) <= 64)) {
    t.bm = 256; t.bn = 64;
  } else {
    t.bm = 128; t.bn = 128;
  }
  if (d->flags & (UDT_GEMM_TRANSPOSED | UDT_GEMM_GEGLU)) { t.bm = 128; t.bn = 128; }
  t.tiles_m = (d->M + t.bm - 1) / t.bm;
  t.tiles_n = (d->N + t.bn - 1) / t.bn;
  t.tiles = t.tiles_m * t.tiles_n * batch;
  t.nkt = d->K / BK;
  t.total = (long long)t.tiles * t.nkt;
  const int slots = resident_slots(d);
  // A parked partial tile costs a 64 KiB slab write + read (~4 K-tiles of operand traffic), so ranges are cut
  // inside tiles only where it buys balance:
  //   few tiles (< slots)            : stream-K, >= 4 K-tiles per workgroup (this is split-K for the deep 8x8 layers)
  //   ragged mid-size, deep K        : stream-K over exactly `slots` workgroups
  //   many tiles (>= 8 x slots)      : persistent, whole tiles per workgroup (tail imbalance < 1/8)
  //   otherwise (shallow K)          : one workgroup per tile
  //   (measured, MI355X round 1: the separate fix-up pass costs ~17 us per launch, so ranges are only cut inside
  //    tiles when there are few tiles AND K is deep, or when the per-workgroup share is >= 160 K-tiles)
  const long long share = (t.total + slots - 1) / slots;
  if (t.tiles < slots / 2 && t.nkt >= 16) {
    long long G = t.total / 8;
    if (G < 1) G = 1;
    if (G > slots) G = slots;
    t.ipw = (int)((t.total + G - 1) / G);
  } else if (t.tiles >= 8 * slots) {
    t.ipw = ((t.tiles + slots - 1) / slots) * t.nkt;
  } else if (share >= 160 && (t.tiles % slots) != 0) {
    t.ipw = (int)share;
  } else {
    t.ipw = t.nkt;
  }
  t.G = (int)((t.total + t.ipw - 1) / t.ipw);
  t.fixup = (t.ipw % t.nkt) != 0;
  return t;
}

template <int BM, int BN, int WGM, int WGN, bool CONV, bool TRANS>
hipError_t launch_cfg(const GemmParams& p, const TilePlan& t, hipStream_t s) {
  constexpr int smem = 2 * (BM + BN) * ROW_BYTES;
  static AttrOnce once;
  auto kern = gemm_kernel<BM, BN, WGM, WGN, CONV, TRANS>;
  hipError_t e = once.ensure(reinterpret_cast<const void*>(kern), smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(t.G), dim3(256), smem, s, p);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (t.fixup) {
    hipLaunchKernelGGL((gemm_fixup_kernel<BM, BN, WGM, WGN, TRANS>), dim3(t.tiles), dim3(256), 0, s, p);
    e = hipGetLastError();
  }
  return e;
}

// ---- 8-wave kernel: host side ------------------------------------------------------------------------------
constexpr size_t G8_HEADER_BYTES = 4096;     // flags[<=1023] + err word, ahead of the slabs

// Tuning knobs (udt_debug_set; every setting gives correct results): kernel generation, tile order, epilogue form.
std::atomic<int> g_n_block{-1};  // tile order of plain GEMMs: -1 automatic (~2 MiB weight blocks), 0 off (M-fastest), n forced
std::atomic<int> g_rows_epi{1};  // row-coalesced (LDS-transposed) epilogues; 0 = direct accumulator-layout stores
#ifdef UDT_MEASURE
std::atomic<int> g_dbg_bits{0};  // cost-attribution modes that switch parts of the finishing code OFF (wrong results):
                                 // compiled only into measurement builds (-DUDT_MEASURE), never into the product library
#endif
constexpr int INTERNAL_DIRECT_EPI = 1 << 30;   // kernel-side flag bit, never part of the public flag set
constexpr int PUBLIC_FLAGS = UDT_GEMM_OUT_F32 | UDT_GEMM_GEGLU | UDT_GEMM_RELU | UDT_GEMM_TRANSPOSED | UDT_GEMM_CONV |
                             UDT_GEMM_SILU_OUT | UDT_GEMM_MX8;
inline int k_tile(const udt_gemm_desc* d) { return (d->flags & UDT_GEMM_MX8) ? 128 : BK; }   // K elements per 128-byte row
inline int elem_bytes(const udt_gemm_desc* d) { return (d->flags & UDT_GEMM_MX8) ? 1 : 2; }

// the 8-wave kernels serve everything the lean family declines, except outputs of <= 64 columns (the UNet's 4-channel output
// convolution): those keep the first-generation 4-wave kernel with its 256 x 64 tile
bool use_gemm8(const udt_gemm_desc* d) {
  if (d->N <= 64) return false;
  const long long ldw = d->ldw > 0 ? d->ldw : d->K;
  // buffer-descriptor addressing uses 31-bit byte offsets per batch element
  if (!(d->flags & UDT_GEMM_CONV) && (long long)d->M * d->lda * elem_bytes(d) >= (1LL << 31)) return false;
  return (long long)d->N * ldw * elem_bytes(d) < (1LL << 31);
}


// Launch a multiple of 8 workgroups so that range_index() can hand every XCD (block b runs on XCD b % 8) one contiguous
// slice of the iteration space: neighbouring tiles then share one L2 (the patch / A rows across N tiles, the weight
// column tile across M tiles, and the stream-K slabs between neighbours).  The padding workgroups own empty ranges and
// exit at once (+1 % images/s in round 2's same-box A/B).
int round_workgroups(int G) { return G > 8 ? ((G + 7) & ~7) : G; }

// Several launch streams in flight (cu_share > 1): what counts is the CU-time a launch consumes, not its latency.  Cutting
// tiles (stream-K) buys latency with CU-time — slab round trips, two prologues per tile, and all partners resident at
// once: measured alone on the chip, 8x32x32 640->640 costs 256 x 90 us of CU-time at full width but 32 x 340 us on 32
// workgroups (tools/bench_cu_share.py).  Whole tiles need no residency (nobody waits), so they queue on whatever CU is
// free.  Used when there are at least ~3/4 as many tiles as the stream's share of workgroup slots.
// cu_share 1 keeps the latency plans.
bool prefer_whole_tiles(const udt_gemm_desc* d, int tiles) {
  if (d->cu_share <= 1) return false;
  return (long long)tiles * 4 >= (long long)(resident_slots(d) / 2) * 3;
}

TilePlan plan_tiles8(const udt_gemm_desc* d) {
  TilePlan t;
  const int batch = d->batch > 0 ? d->batch : 1;
  const bool geglu_or_trans = (d->flags & (UDT_GEMM_TRANSPOSED | UDT_GEMM_GEGLU)) != 0;
  t.bm = 256;
  t.bn = (!geglu_or_trans && d->N % 160 == 0 && d->N % 128 != 0) ? 160 : 128;
  t.tiles_m = (d->M + t.bm - 1) / t.bm;
  t.tiles_n = (d->N + t.bn - 1) / t.bn;
  t.tiles = t.tiles_m * t.tiles_n * batch;
  t.nkt = d->K / k_tile(d);
  t.total = (long long)t.tiles * t.nkt;
  // one 8-wave workgroup per CU.  Whole-tile plans (shallow K, below) have no waits between workgroups, so they may
  // use every CU even when other launch streams share the device (cu_share): excess workgroups simply queue
  const bool whole = t.nkt < 24 || prefer_whole_tiles(d, t.tiles);
  int slots = (whole ? resident_slots_all() : resident_slots(d)) / 2;
  long long G = t.total / 4;                       // >= 4 K-tiles per workgroup
  if (G < 1) G = 1;
  if (G > slots) G = slots;
  t.ipw = (int)((t.total + G - 1) / G);
  // shallow K (< 24 K-tiles): cutting a tile costs more (slab round trip + the finisher's wait) than the imbalance it
  // removes — measured on MI355X: 2048x1280x1280 33.7 -> 28.6 us, 8192x640x640 32.4 -> 20.2 us with whole tiles
  if (whole) t.ipw = ((t.ipw + t.nkt - 1) / t.nkt) * t.nkt;
  t.G = round_workgroups((int)((t.total + t.ipw - 1) / t.ipw));
  t.fixup = (t.ipw % t.nkt) != 0;
  return t;
}

template <int WGM, int WGN, int TM, int TN, bool CONV, bool TRANS, bool STATS = false>
hipError_t launch8(const g8::Params& pp, const TilePlan& t, hipStream_t s) {
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  // the 256x128 configuration transposes its output through LDS: stage 2 plus 16 KiB above the ring (160 KiB total)
  constexpr int smem = (TM == 2 && TN == 2 && !TRANS) ? 160 * 1024 : g8::NSTAGE * (BM + BN) * ROW_BYTES;
  static AttrOnce once;
  auto kern = g8::gemm8_kernel<WGM, WGN, TM, TN, CONV, TRANS, STATS>;
  hipError_t e = once.ensure(reinterpret_cast<const void*>(kern), smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(t.G), dim3(g8::NTHREADS), smem, s, pp);
  return hipGetLastError();
}

// ---- patch-staged 3x3 convolution: host side -------------------------------------------------------------------
// gn: the launch applies GroupNorm on the staged patch (the only variant that reads a second source)
bool conv3p_geometry(const udt_gemm_desc* d, c3p::Geo& ge, bool gn) {
  // since round 4 this 8-wave kernel only serves the launches that apply GroupNorm(+SiLU) on the staged patch
  // (udt_gn_silu_conv3x3_fwd, UDT_FUSE_GN=1): every plain 3x3 / stride 1 convolution runs on the lean / wide kernels, what
  // those decline (maps that are not multiples of 8, fewer than 128 output channels) on the gathered gemm8_kernel<CONV>
  if (!gn) return false;
  if (!(d->flags & UDT_GEMM_CONV) || d->ksize != 3 || d->stride != 1) return false;
  const bool ups = d->upsample != 0;
  if (ups) return false;
  if (d->pad_t != 1 || d->pad_l != 1 || d->N <= 64) return false;
  if (d->Hout != (d->Hin << (ups ? 1 : 0)) || d->Wout != (d->Win << (ups ? 1 : 0))) return false;
  if (d->flags & (UDT_GEMM_GEGLU | UDT_GEMM_TRANSPOSED)) return false;
  if (d->C1 <= 0 || d->C1 % 64 != 0 || d->C2 < 0 || d->C2 % 64 != 0 || (d->C2 != 0 && !gn)) return false;
  const int H = d->Hout, W = d->Wout;              // the tile lives in the OUTPUT map (= input map unless upsampling)
  if (W % 32 == 0 && H % 8 == 0) { ge.TW = 32; ge.TH = 8; ge.NI = 1; }
  else if (W % 16 == 0 && H % 16 == 0) { ge.TW = 16; ge.TH = 16; ge.NI = 1; }      // 16x16, 48x48 (768x768 inputs), ...
  else if (W == 8 && H == 8 && !ups) { ge.TW = 8; ge.TH = 8; ge.NI = 4; }
  else return false;
  ge.B = d->M / (H * W);
  ge.H = d->Hin; ge.W = d->Win; ge.C1 = d->C1; ge.C2 = d->C2; ge.C = d->C1 + d->C2;
  ge.tiles_x = W / ge.TW;
  ge.tiles_y = H / ge.TH;
  ge.img_groups = (ge.B + ge.NI - 1) / ge.NI;
  ge.prow_w = (ups ? ge.TW / 2 : ge.TW) + 2;
  ge.prows_img = ((ups ? ge.TH / 2 : ge.TH) + 2) * ge.prow_w;
  ge.n_pieces = (ge.NI * ge.prows_img + 7) / 8;
  ge.chunks = ge.C / 64;
  // buffer-descriptor addressing: 31-bit byte offsets (bit 31 marks zero padding); pixel indices are kept in 28 bits
  if ((long long)d->M * d->C1 * 2 >= (1LL << 31) || (long long)d->M * d->C2 * 2 >= (1LL << 31) || d->M >= (1 << 28)) return false;
  if ((long long)d->N * (d->ldw > 0 ? d->ldw : d->K) * 2 >= (1LL << 31)) return false;
  if ((long long)ge.B * ge.chunks * 512 >= (1LL << 31)) return false;
  const int bn = (d->N % 160 == 0 && d->N % 128 != 0) ? 160 : 128;
  return ge.n_pieces * 8 <= c3p::patch_rows(bn == 160 ? 5 : 2, gn) && ge.n_pieces >= 8;
}

TilePlan plan_tiles3p(const udt_gemm_desc* d, const c3p::Geo& ge) {
  TilePlan t;
  t.bm = 256;
  t.bn = (d->N % 160 == 0 && d->N % 128 != 0) ? 160 : 128;
  t.tiles_m = ge.img_groups * ge.tiles_y * ge.tiles_x;
  t.tiles_n = (d->N + t.bn - 1) / t.bn;
  t.tiles = t.tiles_m * t.tiles_n;
  t.nkt = ge.chunks;                               // iteration unit of this kernel: one 64-channel chunk (9 K-tiles)
  t.total = (long long)t.tiles * t.nkt;
  const bool whole = t.nkt * 9 < 24 || prefer_whole_tiles(d, t.tiles);
  const int slots = (whole ? resident_slots_all() : resident_slots(d)) / 2;
  long long G = t.total;                           // >= one chunk per workgroup
  if (G > slots) G = slots;
  t.ipw = (int)((t.total + G - 1) / G);
  if (whole) t.ipw = ((t.ipw + t.nkt - 1) / t.nkt) * t.nkt;             // shallow K: whole tiles
  t.G = round_workgroups((int)((t.total + t.ipw - 1) / t.ipw));
  t.fixup = (t.ipw % t.nkt) != 0;
  return t;
}

template <int WGM, int WGN, int TM, int TN, bool GN, bool STATS, bool UPS = false>
hipError_t launch3p(const c3p::CParams& cp, const TilePlan& t, hipStream_t s) {
  constexpr int BN = WGN * TN * 32;
  constexpr int smem = g8::NSTAGE * BN * ROW_BYTES + 2 * c3p::patch_rows(TN, GN) * ROW_BYTES + (GN ? c3p::SCSH_BYTES : 0);
  static_assert(smem <= 160 * 1024, "one workgroup per CU: at most the CU's 160 KiB of LDS");
  static AttrOnce once;
  auto kern = c3p::conv3p_kernel<WGM, WGN, TM, TN, GN, STATS, UPS>;
  hipError_t e = once.ensure(reinterpret_cast<const void*>(kern), smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(t.G), dim3(g8::NTHREADS), smem, s, cp);
  return hipGetLastError();
}

// rows of the output one colstats slot covers for this problem, 0 = statistics not available
struct LeanPlan;
bool lean_stats_probe(const udt_gemm_desc* d, int& rows, int& slots);
int colstats_rows(const udt_gemm_desc* d) {
  if (d->flags & (UDT_GEMM_OUT_F32 | UDT_GEMM_GEGLU | UDT_GEMM_TRANSPOSED)) return 0;
  if ((d->batch > 1) || d->M <= 0 || d->N <= 0 || d->K <= 0 || d->K % BK != 0) return 0;
  int rows = 0, slots = 0;
  if (lean_stats_probe(d, rows, slots)) return rows;             // the launch would take a lean kernel with a STATS epilogue
  c3p::Geo ge;
  if (conv3p_geometry(d, ge, d->in_scsh != nullptr)) {
    rows = (d->N % 160 == 0 && d->N % 128 != 0) ? 32 : 64;
    if ((ge.TW * ge.TH) % rows != 0) return 0;                    // a wave row block lies in one image
  } else if (use_gemm8(d) && g_rows_epi.load(std::memory_order_relaxed)) {
    rows = (plan_tiles8(d).bn == 160) ? 32 : 64;
  } else {
    return 0;
  }
  const int rpb = d->rows_per_batch > 0 ? d->rows_per_batch : d->M;
  return (rpb % rows == 0) ? rows : 0;
}

int colstats_slots(const udt_gemm_desc* d) {
  const int rows = colstats_rows(d);
  if (rows == 0) return 0;
  int lrows = 0, lslots = 0;
  if (lean_stats_probe(d, lrows, lslots)) return lslots;
  c3p::Geo ge;
  if (conv3p_geometry(d, ge, d->in_scsh != nullptr)) return ge.img_groups * ge.tiles_y * ge.tiles_x * (256 / rows);
  return ((d->M + 255) / 256) * (256 / rows);
}


// ---- lean co-resident GEMM family (lean.h): host side --------------------------------------------------------------
// udt_debug_set("lean", v): -1 automatic (default), 0 off, 1 = 4 waves / 128x128 / 2 stages (two workgroups per CU),
// 6 = 8 waves / 256x256 / 2 stages (one per CU) — every setting gives the same results up to fp32 summation order
// (split-K); A/B measurements (tools/bench_gemm_shapes.py).  (Round 3's forced-only 256x128 and three-stage 128x128
// configurations measured slower on every UNet shape and are gone.)
std::atomic<int> g_lean{-1};
std::atomic<int> g_lean_splitk{-2};   // -1 automatic, 1 = never split, n = force n slices where K allows (-2: read UDT_LEAN_SPLITK)
int lean_splitk_knob() {
  int v = g_lean_splitk.load(std::memory_order_relaxed);
  if (v == -2) {
    const char* e = getenv("UDT_LEAN_SPLITK");
    v = e ? atoi(e) : -1;
    g_lean_splitk.store(v, std::memory_order_relaxed);
  }
  return v;
}
// udt_debug_set("share_splitk", v) / UDT_SHARE_SPLITK: 1 (default) the lean GEMM / lean convolution cut K (channel chunks) to fill
// THEIR SHARE of the workgroup slots when the launch shares the device with other streams (udt_gemm_desc.cu_share: the batches in
// flight) — the other lanes fill the CUs a few-tile launch leaves idle, so slices would only add slab round trips (a slice costs
// ~10 us of CU time, 64 KB of fp32 slab written through and read back: round 5's traffic 3.0x the algorithmic bytes); 0: the
// round-5 rule (every launch plans for the whole chip).  A lone launch (cu_share 1) is planned as before.
std::atomic<int> g_share_splitk{-2};
int share_splitk() {
  int v = g_share_splitk.load(std::memory_order_relaxed);
  if (v == -2) {
    const char* e = getenv("UDT_SHARE_SPLITK");
    v = e ? (atoi(e) != 0) : 1;
    g_share_splitk.store(v, std::memory_order_relaxed);
  }
  return v;
}
int share_of(const udt_gemm_desc* d) { return (share_splitk() && d->cu_share > 1) ? d->cu_share : 1; }
bool lean_stats_enabled() { return true; }     // statistics-emitting launches run on the lean kernels wherever a lean plan exists

struct LeanPlan {
  int stats_rows = 0;      // rows per colstats slot when the launch emits statistics (lean.h wave_colstats), else 0
  int cfg;                 // 1, 6 as above; 5 = 4 waves / 128x160 / 2 stages (N = 320, 960); 7 = row-resident (rowres.h: tiles_n =
                           // column splits, kt_per = 64-column chunks per split)
  int bm, bn, nw, smem;
  int tiles_m, tiles_n, tiles, nkt, splitk, kt_per, G, n_block;
};

int lean_mode() {
  int v = g_lean.load(std::memory_order_relaxed);
  if (v == -1) {
    const char* e = getenv("UDT_LEAN");
    v = e ? atoi(e) : -2;                                  // -2: automatic
    g_lean.store(v, std::memory_order_relaxed);
  }
  return v;
}

// udt_debug_set("rowres", v): 1 (default) the row-resident kernel where it applies, 0 never (same-process A/B, tools/ab_step.py;
// a tuning knob of the debug interface like "lean_splitk", not an environment switch)
std::atomic<int> g_rowres{-2};
bool rowres_on() {
  int v = g_rowres.load(std::memory_order_relaxed);
  if (v == -2) {
    const char* e = getenv("UDT_ROWRES");                  // (A/B of whole bench runs; udt_debug_set("rowres") for same-process A/Bs)
    v = e ? (atoi(e) != 0) : 1;
    g_rowres.store(v, std::memory_order_relaxed);
  }
  return v != 0;
}
// UDT_LEAN256_LANES (default 1): the 256 x 256 / one-workgroup-per-CU tile also for launches that share the device with other
// streams (0: under lanes the co-resident 128 x 128 tile; round 6 A/B, profiles/r06_ab_one_per_cu_under_lanes.txt)
std::atomic<int> g_lean256_lanes{-2};
bool lean256_under_lanes() {
  int v = g_lean256_lanes.load(std::memory_order_relaxed);
  if (v == -2) {
    const char* e = getenv("UDT_LEAN256_LANES");
    v = e ? (atoi(e) != 0) : 1;
    g_lean256_lanes.store(v, std::memory_order_relaxed);
  }
  return v != 0;
}

// udt_debug_set("conv_n4", v) / UDT_CONV_N4: 1 (default) the dot-product kernel of conv_n4.h for 3x3 convolutions with four output
// channels and fp32 output, 0: the gathered MFMA kernel on a 64-column tile
std::atomic<int> g_conv_n4{-2};
bool conv_n4_applies(const udt_gemm_desc* d) {
  int v = g_conv_n4.load(std::memory_order_relaxed);
  if (v == -2) {
    const char* e = getenv("UDT_CONV_N4");
    v = e ? (atoi(e) != 0) : 1;
    g_conv_n4.store(v, std::memory_order_relaxed);
  }
  if (!v) return false;
  if (d->flags != (UDT_GEMM_CONV | UDT_GEMM_OUT_F32) || d->ksize != 3 || d->stride != 1 || d->pad_t != 1 || d->pad_l != 1 || d->upsample)
    return false;
  if (d->N != 4 || d->C2 != 0 || d->C1 % 64 != 0 || d->C1 > 512 || d->Hout != d->Hin || d->Wout != d->Win || d->ldo % 4 != 0) return false;
  if (d->residual || d->rowvec || d->in_scsh || d->colstats || d->colscale || d->batch > 1 || d->alpha != 1.0f) return false;
  if ((d->ldw > 0 ? d->ldw : d->K) % 8 != 0) return false;
  if ((reinterpret_cast<uintptr_t>(d->a) | reinterpret_cast<uintptr_t>(d->w) | reinterpret_cast<uintptr_t>(d->out)) & 15) return false;
  return true;
}

bool lean_plan(const udt_gemm_desc* d, LeanPlan& t, bool want_stats) {
  const int mode = lean_mode();
  if (mode == 0) return false;
  constexpr int unsupported = UDT_GEMM_OUT_F32 | UDT_GEMM_RELU | UDT_GEMM_TRANSPOSED | UDT_GEMM_SILU_OUT;
  if (d->flags & unsupported) return false;
  // 1x1 / stride-1 convolutions are plain GEMMs over the pixels; two NHWC sources (the UNet's skip concat in front of a
  // ResBlock's skip_connection, reference openaimodel.py:218-231,620) = two A sources along K
  const bool conv1 = (d->flags & UDT_GEMM_CONV) != 0;
  if (conv1 && (d->ksize != 1 || d->stride != 1 || d->upsample || d->pad_t != 0 || d->pad_l != 0 || d->Hout != d->Hin ||
                d->Wout != d->Win || d->C1 % BK != 0 || d->C2 % BK != 0 || d->ln_colsum || (d->flags & UDT_GEMM_GEGLU)))
    return false;
  const bool mx8 = (d->flags & UDT_GEMM_MX8) != 0;          // e4m3 operands, MX block scales on A (lean.h FP8)
  const bool emit = d->q8_out != nullptr;                   // the result again as an MX8 activation (lean.h EMIT)
  if (d->batch > 1 || d->in_scsh || (d->colscale != nullptr) != mx8) return false;
  const bool geglu = (d->flags & UDT_GEMM_GEGLU) != 0;
  const bool ln = d->ln_colsum != nullptr;
  if (want_stats && (geglu || ln || !lean_stats_enabled())) return false;
  if (mx8 || emit) {
    // instances (lean.hip launch_lean_mx8): bf16 plain + emit; fp8 plain (+ emit | + statistics), fp8 LayerNorm-folded plain,
    // fp8 LayerNorm-folded GEGLU + emit — all on the 128 x 128 configuration
    if (conv1 || d->rowvec || mode == 0 || (mode > 0 && mode != 1)) return false;
    if (mx8 && (!d->a_scale || d->K % 128 != 0 || d->lda % 16 != 0 || (d->ldw > 0 && d->ldw % 16 != 0))) return false;
    if (mx8 && ln && (!d->rowstat_in || d->rowstat_in_parts <= 0)) return false;
    if (!mx8 && geglu) return false;
    if (mx8 && geglu && !(ln && emit)) return false;
    if (mx8 && !geglu && ln && emit && d->rowstat_out) return false;
    if (emit && (want_stats || !d->q8_scale || d->ld_q8 % 8 != 0 || d->N % (geglu ? 64 : 32) != 0)) return false;
    if (d->rowstat_out && (!emit || geglu)) return false;
    // (columns from q8_fixed_col on leave with a fixed multiplier and the unit scale byte: whole 32-column blocks, no GEGLU halves)
    if (d->q8_fixed_col > 0 && (!emit || geglu || d->q8_fixed_col % 32 != 0 || !(d->q8_fixed_mul > 0.f))) return false;
    if ((reinterpret_cast<uintptr_t>(d->q8_out) | reinterpret_cast<uintptr_t>(d->colscale)) & 15) return false;
  } else if (d->rowstat_out || d->rowstat_in) {
    return false;
  }
  const int KT = mx8 ? 128 : BK;                            // K elements per 128-byte LDS row
  if (d->N <= 64 || d->N % 8 != 0 || d->K % KT != 0 || (!conv1 && d->lda % 8 != 0) || (d->ldo % 8 != 0 && d->out)) return false;
  if (d->residual && d->ldr % 8 != 0) return false;
  if (geglu && (d->N % 64 != 0 || d->residual || d->rowvec)) return false;
  const long long ldw = d->ldw > 0 ? d->ldw : d->K;
  const long long lda_eff = conv1 ? (d->C1 > d->C2 ? d->C1 : d->C2) : d->lda;
  if ((long long)d->M * lda_eff * 2 >= (1LL << 31) || (long long)d->N * ldw * 2 >= (1LL << 31)) return false;
  if (d->M >= (1 << 28)) return false;                      // (MX8 scale / statistics indices)
  if ((reinterpret_cast<uintptr_t>(d->out) | reinterpret_cast<uintptr_t>(d->residual) | reinterpret_cast<uintptr_t>(d->bias) |
       reinterpret_cast<uintptr_t>(d->rowvec) | reinterpret_cast<uintptr_t>(d->ln_colsum)) & 15) return false;
  if (d->rowvec && ((d->ld_rowvec > 0 ? d->ld_rowvec : d->N) % 4 != 0)) return false;
  // forced configurations: 1 and 6 only (5 = the 128 x 160 tile is chosen here, for plain N = 320 / 960 outputs: its wave
  // grid has no GEGLU form); anything else falls back to the 8-wave kernels instead of silently taking a default instance
  if (mode > 0 && mode != 1 && mode != 6 && mode != 7) return false;
  // 7 = rowres.h: the LayerNorm-folded projections with K = 320 and enough rows to give every CU a 256-row block: the rows'
  // A fragments stay in registers, the weights stream through LDS in 64-row chunks; automatic where it applies
  if (!mx8 && (!emit || (ln && !geglu && !d->rowstat_out)) && (mode == 7 || (mode < 0 && rowres_on()))) {
    // (its A fragments are 16-byte vector loads straight from global memory: a 16-byte aligned base; lda % 8 == 0 is checked above)
    // (its emitting epilogue addresses through 32-bit buffer offsets and switches to the fixed multiplier per 64-column chunk)
    const bool emit_fits = !emit || ((d->q8_fixed_col <= 0 || d->q8_fixed_col % 64 == 0) && (long long)d->M * d->ld_q8 < (1LL << 31) &&
                                     (long long)d->M * (d->out ? d->ldo : 0) * 2 < (1LL << 31) && (long long)((d->N + 127) / 128) * d->M * 4 < (1LL << 31));
    if (ln && !want_stats && d->K == 320 && d->N % 64 == 0 && !d->residual && !d->rowvec && !conv1 && emit_fits &&
        (reinterpret_cast<uintptr_t>(d->a) & 15) == 0) {
      const int tiles_m = (d->M + 255) / 256, chunks = d->N / 64;
      int ns = (device_cus() + tiles_m - 1) / tiles_m;           // column splits: one workgroup per CU ...
      if (ns > chunks / 4) ns = chunks / 4;                       // ... of at least 4 chunks (the A load amortised)
      const int cmax = geglu ? 20 : 16;                           // (column constants of <= 20 / 16 chunks in LDS, rowres.h)
      const int ns_min = (chunks + cmax - 1) / cmax;
      if (ns < ns_min) ns = ns_min;
      // automatic where measured faster than the tiled kernels (profiles/r04_rowres.txt): GEGLU from ~3/4 of the CUs busy
      // (16384 rows: 48 vs 59 us; 32768: 90 vs 103 us), the plain epilogue only with every CU busy (32768 x 960: 36.6 vs 38.2 us;
      // 16384 rows: slower)
      const long long wgs = (long long)tiles_m * ns;
      // (the bf16 LayerNorm-folded emitting epilogue exists on this kernel only: taken at any size)
      if (ns >= 1 && (mode == 7 || emit || (geglu ? wgs * 4 >= 3LL * device_cus() : wgs >= device_cus()))) {
        t.cfg = 7; t.stats_rows = 0; t.nw = 4; t.bm = 256; t.bn = 64; t.smem = 0;
        t.tiles_m = tiles_m; t.kt_per = (chunks + ns - 1) / ns; t.tiles_n = (chunks + t.kt_per - 1) / t.kt_per;
        t.tiles = t.tiles_m * t.tiles_n; t.nkt = d->K / BK; t.splitk = 1; t.n_block = 1;
        t.G = round_workgroups(t.tiles);
        return true;
      }
    }
    if (mode == 7 || (emit && ln)) return false;
  }
  t.cfg = (mode > 0) ? mode : 1;
  if (!mx8 && !emit) {
  if (!geglu && d->N % 160 == 0 && d->N % 128 != 0 && t.cfg == 1) t.cfg = 5;
  // many tiles and a wide output: the 256 x 256 tile (8 waves, one workgroup per CU) halves the LDS-DMA instructions per MFMA.
  // Measured (profiles/r03_gemm_shapes_256.txt): faster from ~2 tiles per CU up (32768x2560x320 GEGLU 93 -> 81 us,
  // 32768x960x320 33 -> 32 us), slower below (every M <= 2048 shape)
  if (mode <= 0 && d->N >= 768 && (long long)((d->M + 255) / 256) * ((d->N + 255) / 256) >= 2LL * device_cus() &&
      (d->cu_share <= 1 || lean256_under_lanes())) t.cfg = 6;
  }
  t.stats_rows = 0;
  if (want_stats) {
    // statistics-emitting epilogues exist for the two 4-wave plain kernels; a slot = one wave row block, inside one sample
    if (t.cfg != 1 && t.cfg != 5) {
      if (mode > 0) return false;
      t.cfg = (d->N % 160 == 0 && d->N % 128 != 0) ? 5 : 1;
    }
    if (mx8) t.cfg = 1;
    t.stats_rows = t.cfg == 1 ? 64 : 32;
    const int rpb = d->rows_per_batch > 0 ? d->rows_per_batch : d->M;
    if (rpb % t.stats_rows != 0) return false;
  }
  switch (t.cfg) {
    case 1: t.nw = 4; t.bm = 128; t.bn = 128; t.smem = 2 * (128 + 128) * ROW_BYTES; break;
    case 5: t.nw = 4; t.bm = 128; t.bn = 160; t.smem = 128 * 160 * 4; break;          // the fp32 staging rows exceed the ring
    case 6: t.nw = 8; t.bm = 256; t.bn = 256; t.smem = 2 * (256 + 256) * ROW_BYTES; break;   // one per CU: 128x64 per wave
    default: return false;
  }
  {
    // the epilogue's fp32 staging rows re-use the ring (+ [BM][mean, rstd] behind them when two wave columns share rows)
    const int epi = (t.cfg == 6 ? t.bm * t.bn : t.bm * t.bn * 4) + ((ln && t.cfg != 5) ? t.bm * 8 : 0);   // (cfg 6: passes of 32 rows per wave)
    if (t.smem < epi) t.smem = epi;
  }
  t.tiles_m = (d->M + t.bm - 1) / t.bm;
  t.tiles_n = (d->N + t.bn - 1) / t.bn;
  t.tiles = t.tiles_m * t.tiles_n;
  t.nkt = d->K / KT;
  // split K when the tiles alone leave most of the chip idle: slices of >= 4 K-tiles, ~1.5 units per workgroup slot
  int slots = device_cus() * ((t.cfg == 1 || t.cfg == 5) ? 2 : 1) / share_of(d);
  if (slots < 1) slots = 1;
  int sk = 1;
  const int knob = lean_splitk_knob();
  if (!ln && t.tiles <= 1023) {
    if (knob > 1) sk = knob;
    // measured (profiles/r03_gemm_shapes_lean_splitk.txt): a slice costs ~10 us of slab round trip + ticket, so K is cut
    // only when it is deep (>= 40 K-tiles) and the tiles leave more than half of the workgroup slots idle
    else if (knob < 0 && t.tiles * 2 <= slots && t.nkt >= (mx8 ? 20 : 40)) { sk = slots / t.tiles; if (sk > t.nkt / (mx8 ? 8 : 16)) sk = t.nkt / (mx8 ? 8 : 16); }
    if (sk > t.nkt / 4) sk = t.nkt / 4;
    while (sk > 1 && (long long)t.tiles * sk * t.bm * t.bn * 4 > (64LL << 20)) --sk;    // slabs stay inside the workspace
    if (sk < 1) sk = 1;
  }
  t.kt_per = (t.nkt + sk - 1) / sk;
  t.splitk = (t.nkt + t.kt_per - 1) / t.kt_per;
  const int units = t.tiles * t.splitk;
  t.G = round_workgroups(units);
  // tile order: an XCD's slice of ~64 concurrently resident tiles should be ~8 M-tiles x 8 N-tiles
  int nb = g_n_block.load(std::memory_order_relaxed);
  if (nb <= 0) nb = (t.tiles_m <= 8) ? t.tiles_n : 8;
  if (nb > t.tiles_n) nb = t.tiles_n;
  t.n_block = nb;
  return true;
}

size_t lean_workspace(const LeanPlan& t) {
  return t.splitk > 1 ? G8_HEADER_BYTES + (size_t)t.tiles * t.splitk * t.bm * t.bn * sizeof(float) : 0;
}

// ---- lean 3x3 convolution (lean.h lconv3_kernel): host side ------------------------------------------------------------
// udt_debug_set("lean_conv", v): -1 automatic (default: on), 0 off, 1 on
std::atomic<int> g_lean_conv{-1};
// udt_debug_set("wide_conv", v): -1 automatic (default: where a launch fills the CUs with 256-pixel x 160-channel tiles),
// 0 off, 1 wherever the geometry allows (wide.h wconv3_kernel)
std::atomic<int> g_wide_conv{-2};
int wide_conv_mode() {
  int v = g_wide_conv.load(std::memory_order_relaxed);
  if (v == -2) {
    const char* e = getenv("UDT_WIDE_CONV");
    v = e ? atoi(e) : -1;
    g_wide_conv.store(v, std::memory_order_relaxed);
  }
  return v;
}
#ifdef UDT_MEASURE
std::atomic<int> g_lconv_dbg{0};   // cost attribution of the lean convolution's loop (C3Params.dbg): wrong results
#endif

// udt_debug_set("wide_lanes_eff", percent) / UDT_WIDE_LANES_EFF: the share of its CUs a launch must fill to take the wide convolution
// when three or more launch streams share the device (default 70; 85 = the rule of a lone launch)
std::atomic<int> g_wide_lanes_eff{-2};
double wide_lanes_eff() {
  int v = g_wide_lanes_eff.load(std::memory_order_relaxed);
  if (v == -2) {
    const char* e = getenv("UDT_WIDE_LANES_EFF");
    v = e ? atoi(e) : 70;
    g_wide_lanes_eff.store(v, std::memory_order_relaxed);
  }
  return v / 100.0;
}

// channel-chunk slices per tile of the lean / wide convolution (ticket split-K): ONE rule for the plan and for the wide kernel's
// eligibility test (which used to predict chunks / 5 and ignore the forced knob while the plan cut chunks / 4)
int conv_chunk_slices(long long tiles, int slots, int chunks, int knob) {
  long long sk = 1;
  if (tiles <= 1023) {
    if (knob > 1) { sk = knob; if (sk > chunks) sk = chunks; }              // forced (tests / A-B): any cut the chunks allow
    // (slices of >= 4 chunks = 36 taps; round 4: was >= 5 — 8 x 8 maps, 1280 -> 1280 on 8 samples: 5 slices 29.2 us, 4 slices 32.1 us,
    //  profiles/r04_pair_conv_rejected.txt)
    else if (knob < 0 && tiles * 2 <= slots && chunks >= 10) { sk = slots / tiles; if (sk > chunks / 4) sk = chunks / 4; }
    if (sk < 1) sk = 1;
  }
  return (int)sk;
}

bool lean_conv_plan(const udt_gemm_desc* d, lg::C3Params& c, bool want_stats) {
  int on = g_lean_conv.load(std::memory_order_relaxed);
  if (on < 0) {
    const char* e = getenv("UDT_LEAN_CONV");
    on = (e && e[0] == '0') ? 0 : 1;
    g_lean_conv.store(on, std::memory_order_relaxed);
  }
  if (!on || lean_mode() == 0) return false;
  if (d->flags != UDT_GEMM_CONV || d->ksize != 3 || d->stride != 1 || d->pad_t != 1 || d->pad_l != 1) return false;
  if (d->C2 != 0 || d->in_scsh || d->colscale || d->batch > 1) return false;
  if (want_stats && !lean_stats_enabled()) return false;
  if (d->N < 128 || d->N % 8 != 0 || d->C1 <= 0 || d->C1 % 64 != 0) return false;
  const bool ups = d->upsample != 0;
  if (d->Hout != (d->Hin << (ups ? 1 : 0)) || d->Wout != (d->Win << (ups ? 1 : 0))) return false;
  // pixel tile: 16 x 8 (two MFMA row tiles per wave), or 8 x 8 for the small maps (8 x 8, 24 x 24, ...); 16 x 16 pixels x 160
  // channels on the wide kernel (wide.h: one workgroup per CU) when its tiles — cut into channel-chunk slices where needed —
  // give every CU a unit
  c.bn = 128; c.wgm = 2;
  bool wide = false;
  int wide_slots = device_cus();
  {
    // measured (tools/bench_wide_conv.py, profiles/r04_wide_conv.txt): 1.25-1.3x at the 64x64 level (256 tiles of a UNet call on 8
    // samples: one per CU, no half-empty third channel tile), 1.03-1.2x at 32x32 with two channel-chunk slices per tile, slower
    // where the tiles need four slices (16x16 level: the slab round trip) or fill the last round of workgroups badly.  A launch
    // that shares the device with s - 1 other streams (cu_share) aims at its share of the CUs: the two CFG halves of a lone batch
    // run 128 tiles each, side by side
    const int wm = wide_conv_mode();
    if (wm != 0 && !ups && d->N % 160 == 0 && d->Wout % 16 == 0 && d->Hout % 16 == 0) {
      const int cus = device_cus(), share = d->cu_share > 1 ? d->cu_share : 1;
      wide_slots = cus / share > 0 ? cus / share : 1;
      const long long wt = (long long)(d->M / 256) * (d->N / 160);
      const long long sk = conv_chunk_slices(wt, wide_slots, d->C1 / 64, lean_splitk_knob());
      const long long units = wt * sk;
      const long long rounds = (units + cus - 1) / cus;
      const double eff = units > cus ? (double)units / (double)(rounds * cus) : ((double)units * share >= cus ? 1.0 : (double)units * share / cus);
      // round 6: with three or more batches in flight a launch that fills only 3/4 of its share still goes wide — the 16 x 16 level's
      // 64 whole tiles (one per CU for all 20 channel chunks, no slab exchange) against 160 lean workgroups: +1.2 % images/s
      // (profiles/r06_ab_wide_conv_16x16_under_lanes.txt); alone (four slices per tile) the lean kernel stays 3 us ahead
      wide = (wm > 0) || (sk <= 2 && eff >= ((share >= 3 && share_of(d) >= 3) ? wide_lanes_eff() : 0.85));
    }
  }
  if (wide) { c.geo = 3; c.tw = 16; c.th = 16; c.bn = 160; c.wgm = 4; }
  else if (d->Wout % 16 == 0 && d->Hout % 8 == 0) { c.geo = ups ? 2 : 0; c.tw = 16; c.th = 8; }
  else if (!ups && d->Wout % 8 == 0 && d->Hout % 8 == 0) { c.geo = 1; c.tw = 8; c.th = 8; }
  else return false;
  if (d->ldo % 8 != 0 || (d->residual && d->ldr % 8 != 0)) return false;
  const long long ldw = d->ldw > 0 ? d->ldw : d->K;
  if ((long long)d->M * d->C1 * 2 >= (1LL << 31) || (long long)d->N * ldw * 2 >= (1LL << 31)) return false;
  if ((reinterpret_cast<uintptr_t>(d->out) | reinterpret_cast<uintptr_t>(d->residual) | reinterpret_cast<uintptr_t>(d->bias) |
       reinterpret_cast<uintptr_t>(d->rowvec)) & 15) return false;
  if (d->rowvec && ((d->ld_rowvec > 0 ? d->ld_rowvec : d->N) % 4 != 0)) return false;
  c.dbg = 0;
#ifdef UDT_MEASURE
  c.dbg = g_lconv_dbg.load(std::memory_order_relaxed);
#endif
  c.N = d->N; c.C = d->C1; c.H = d->Hin; c.W = d->Win; c.B = d->M / (d->Hout * d->Wout);
  c.ldw = (int)ldw; c.ldo = d->ldo; c.ldr = d->ldr; c.ldrv = d->ld_rowvec > 0 ? d->ld_rowvec : d->N;
  c.alpha = d->alpha;
  c.tiles_x = d->Wout / c.tw; c.tiles_y = d->Hout / c.th;
  c.tiles_m = c.B * c.tiles_x * c.tiles_y;
  c.tiles_n = (d->N + c.bn - 1) / c.bn;
  c.tiles = c.tiles_m * c.tiles_n;
  c.chunks = c.C / 64;
  const int slots = c.geo == 3 ? wide_slots : (2 * device_cus() / share_of(d) > 0 ? 2 * device_cus() / share_of(d) : 1);
  int sk = conv_chunk_slices(c.tiles, slots, c.chunks, lean_splitk_knob());
  while (sk > 1 && (long long)c.tiles * sk * c.tw * c.th * c.bn * 4 > (64LL << 20)) --sk;       // slabs stay inside the workspace
  c.ch_per = (c.chunks + sk - 1) / sk;
  c.splitk = (c.chunks + c.ch_per - 1) / c.ch_per;
  c.G = round_workgroups(c.tiles * c.splitk);
  // tile order: ~64 concurrently resident tiles per XCD; n_block weight tiles per patch (conv3p's rule)
  {
    const double patch = (double)((ups ? c.tw / 2 : c.tw) + 2) * ((ups ? c.th / 2 : c.th) + 2) * c.C * 2.0, wtile = 9.0 * c.bn * c.C * 2.0;
    int nb = g_n_block.load(std::memory_order_relaxed);
    if (nb <= 0) nb = (int)(std::sqrt(64.0 * patch / wtile) + 0.5);
    if (nb < 1) nb = 1;
    if (nb > c.tiles_n) nb = c.tiles_n;
    c.n_block = nb;
  }
  c.a_bytes = (unsigned)((long long)c.B * d->Hin * d->Win * d->C1 * 2);
  c.w_bytes = (unsigned)((long long)d->N * ldw * 2);
  c.counters = nullptr; c.slabs = nullptr;
  return true;
}

size_t lean_conv_workspace(const lg::C3Params& c) {
  return c.splitk > 1 ? G8_HEADER_BYTES + (size_t)c.tiles * c.splitk * c.tw * c.th * c.bn * sizeof(float) : 0;
}

// would this problem run on a lean kernel WITH a statistics-emitting epilogue?  (udt_gemm_colstats_rows / _slots, asked by
// the caller before it allocates the statistics and sets udt_gemm_desc.colstats)
bool lean_stats_probe(const udt_gemm_desc* d, int& rows, int& slots) {
  if (d->in_scsh) return false;
  {
    lg::C3Params c3;
    if (lean_conv_plan(d, c3, true)) {
      rows = c3.tw * c3.th / c3.wgm;                              // one slot per wave pixel block of a tile
      slots = c3.tiles_m * c3.wgm;
      return true;
    }
  }
  if ((d->flags & UDT_GEMM_CONV) && d->ksize == 3) return false;
  LeanPlan lt;
  if (lean_plan(d, lt, true)) {
    rows = lt.stats_rows;
    slots = lt.tiles_m * (lt.bm / lt.stats_rows);
    return true;
  }
  return false;
}
}  // namespace

extern "C" int udt_debug_set(const char* key, int32_t value) {
  if (!key) return UDT_ERR_BAD_ARG;
  if (!strcmp(key, "n_block")) { g_n_block.store(value); return UDT_OK; }
  if (!strcmp(key, "rows_epi")) { g_rows_epi.store(value ? 1 : 0); return UDT_OK; }
  if (!strcmp(key, "lean")) { g_lean.store(value < 0 ? -2 : value); return UDT_OK; }
  if (!strcmp(key, "rowres")) { g_rowres.store(value < 0 ? 1 : (value ? 1 : 0)); return UDT_OK; }
  if (!strcmp(key, "wide_lanes_eff")) { g_wide_lanes_eff.store(value < 0 ? -2 : value); return UDT_OK; }
  if (!strcmp(key, "conv_n4")) { g_conv_n4.store(value < 0 ? -2 : (value ? 1 : 0)); return UDT_OK; }
  if (!strcmp(key, "lean256_lanes")) { g_lean256_lanes.store(value < 0 ? -2 : (value ? 1 : 0)); return UDT_OK; }
  if (!strcmp(key, "lean_splitk")) { g_lean_splitk.store(value); return UDT_OK; }
  if (!strcmp(key, "share_splitk")) { g_share_splitk.store(value < 0 ? -2 : (value ? 1 : 0)); return UDT_OK; }
  if (!strcmp(key, "lean_conv")) { g_lean_conv.store(value < 0 ? -1 : (value ? 1 : 0)); return UDT_OK; }
  if (!strcmp(key, "wide_conv")) { g_wide_conv.store(value < 0 ? -1 : (value ? 1 : 0)); return UDT_OK; }
#ifdef UDT_MEASURE
  if (!strcmp(key, "lconv_dbg")) { g_lconv_dbg.store(value); return UDT_OK; }
  static const struct { const char* k; int bit; } bits[] = {{"no_xchg", 28}, {"no_epi", 27}, {"no_store", 26}, {"no_res", 25},
                                                            {"no_bias", 24}, {"no_fast", 22}};
  for (const auto& b : bits)
    if (!strcmp(key, b.k)) {
      int cur = g_dbg_bits.load();
      g_dbg_bits.store((cur & ~(1 << b.bit)) | (value ? (1 << b.bit) : 0));
      return UDT_OK;
    }
#endif
  return UDT_ERR_BAD_ARG;
}

// Stream-K kernels that time out waiting for a partner workgroup (a launch that was not co-resident: wrong cu_share,
// foreign long-running kernels holding the CUs) poison their output tile with NaN, leave the flags alone and set the
// workspace's err word.  Synchronises `stream`, reads the word back and, if set, re-zeroes the header so the workspace
// is usable again.  Call at natural sync points (end of a sampling loop, tests).
extern "C" int udt_check_async_error(void* workspace, size_t workspace_bytes, void* stream) {
  if (!workspace || workspace_bytes < G8_HEADER_BYTES) return UDT_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int err = 0;
  hipError_t e = hipMemcpyAsync(&err, reinterpret_cast<int*>(workspace) + 1023, sizeof(int), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) return udt_set_hip_error(e);
  if (err == 0) return UDT_OK;
  e = hipMemsetAsync(workspace, 0, G8_HEADER_BYTES, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) return udt_set_hip_error(e);
  return UDT_ERR_ASYNC;
}

extern "C" int32_t udt_gemm_colstats_rows(const udt_gemm_desc* d) { return d ? colstats_rows(d) : 0; }
extern "C" int32_t udt_gemm_colstats_slots(const udt_gemm_desc* d) { return d ? colstats_slots(d) : 0; }
extern "C" int32_t udt_gemm_q8_ok(const udt_gemm_desc* d) {
  if (!d || !d->q8_out || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
  LeanPlan lt;
  return lean_plan(d, lt, d->colstats != nullptr) ? 1 : 0;
}

extern "C" int32_t udt_gemm_rowstat_parts(const udt_gemm_desc* d) {
  if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
  // the MX8-emitting epilogues live on the 128 x 128 lean configuration: two wave columns of 64 -> one part per 64 output columns
  udt_gemm_desc t = *d;
  alignas(16) static const uint64_t probe_dst[2] = {0, 0};              // (plan-only: a 16-byte aligned stand-in for q8_out / q8_scale)
  if (!t.q8_out) { t.q8_out = const_cast<uint64_t*>(probe_dst); t.q8_scale = const_cast<uint64_t*>(probe_dst); if (t.ld_q8 <= 0) t.ld_q8 = (t.N + 15) / 16 * 16; }
  LeanPlan lt;
  if (!lean_plan(&t, lt, false) || lt.cfg != 1 || (d->flags & UDT_GEMM_GEGLU)) return 0;
  return (d->N + 63) / 64;
}
extern "C" int32_t udt_gemm_in_scsh_ok(const udt_gemm_desc* d) {
  if (!d || d->K <= 0 || d->K % BK != 0) return 0;
  c3p::Geo ge;
  return conv3p_geometry(d, ge, true) ? 1 : 0;
}

extern "C" int udt_gn_silu_conv3x3_fwd(const udt_gemm_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
  if (!d || !(d->flags & UDT_GEMM_CONV) || d->ksize != 3 || d->stride != 1 || d->pad_t != 1 || d->pad_l != 1 || !d->in_scsh)
    return UDT_ERR_BAD_ARG;
  return udt_gemm(d, workspace, workspace_bytes, stream);
}

extern "C" int udt_ln_gemm_fwd(const udt_gemm_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
  if (!d || !d->ln_colsum || !d->bias || !(d->ln_eps > 0.f)) return UDT_ERR_BAD_ARG;
  LeanPlan lt;
  if (!lean_plan(d, lt, false)) return UDT_ERR_BAD_SHAPE;
  return udt_gemm(d, workspace, workspace_bytes, stream);
}

extern "C" size_t udt_gemm_workspace_bytes(const udt_gemm_desc* d) {
  if (!d || d->K <= 0 || d->M <= 0 || d->N <= 0 || d->K % k_tile(d) != 0) return 0;
  if (conv_n4_applies(d)) return 0;
  {
    LeanPlan lt;
    if (lean_plan(d, lt, d->colstats != nullptr)) return lean_workspace(lt);
    lg::C3Params c3;
    if (lean_conv_plan(d, c3, d->colstats != nullptr)) return lean_conv_workspace(c3);
  }
  {
    c3p::Geo ge;
    if (conv3p_geometry(d, ge, d->in_scsh != nullptr)) {
      TilePlan t3 = plan_tiles3p(d, ge);
      if (!t3.fixup) return 0;
      return G8_HEADER_BYTES + (size_t)t3.G * t3.bm * t3.bn * sizeof(float);
    }
  }
  if (use_gemm8(d)) {
    TilePlan t8 = plan_tiles8(d);
    if (!t8.fixup) return 0;
    return G8_HEADER_BYTES + (size_t)t8.G * t8.bm * t8.bn * sizeof(float);
  }
  TilePlan t = plan_tiles(d);
  // (the first-generation kernel survives in its 256 x 64 geometry only: a problem it would have taken on 128 x 128 tiles — one the
  //  newer kernels decline, e.g. operands of 2 GiB and more — is refused by udt_gemm with UDT_ERR_BAD_SHAPE; no workspace to plan)
  if (t.bm != 256 || !t.fixup) return 0;
  // the first G8_HEADER_BYTES of the workspace hold the 8-wave kernels' flags and are never used for slabs
  return G8_HEADER_BYTES + (size_t)t.G * 2 * SLAB_FLOATS * sizeof(float);
}

extern "C" int udt_gemm(const udt_gemm_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
  if (!d || !d->a || !d->w) return UDT_ERR_BAD_ARG;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return UDT_ERR_BAD_SHAPE;
  if (d->K % k_tile(d) != 0) return UDT_ERR_BAD_SHAPE;
  if (d->N % 4 != 0) return UDT_ERR_BAD_SHAPE;
  const bool mx8 = (d->flags & UDT_GEMM_MX8) != 0;
  if (mx8 || d->q8_out || d->rowstat_out || d->rowstat_in) {
    // MX8 operands / the MX8-emitting epilogues exist on the lean 128 x 128 kernels only (lean_plan): say so instead of falling
    // through to a kernel that would read e4m3 bytes as bf16
    LeanPlan lt8;
    if (!lean_plan(d, lt8, d->colstats != nullptr)) return UDT_ERR_BAD_ARG;
  } else if (d->colscale) {
    return UDT_ERR_BAD_ARG;
  }
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
  if (!d->out && !d->q8_out) return UDT_ERR_BAD_ARG;
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
  if (conv_n4_applies(d)) {
    // four output channels, fp32 output (the UNet's `out` convolution, the VAE decoder's conv_out): the dot-product kernel of conv_n4.h
    cn4::Params cp;
    cp.a = reinterpret_cast<const uint16_t*>(d->a); cp.w = reinterpret_cast<const uint16_t*>(d->w); cp.bias = d->bias;
    cp.out = reinterpret_cast<float*>(d->out);
    cp.B = d->M / (d->Hout * d->Wout); cp.H = d->Hin; cp.W = d->Win; cp.C = d->C1;
    cp.ldw = d->ldw > 0 ? d->ldw : d->K; cp.ldo = d->ldo;
    cp.tiles_x = (d->Win + 7) / 8; cp.tiles_y = (d->Hin + 7) / 8;
    const size_t smem = (size_t)100 * (d->C1 + 8) * 2 + (size_t)4 * 9 * d->C1 * 2;      // halo + weights (>= the 4 KiB of the reduction)
    static AttrOnce once;
    hipError_t ea = once.ensure(reinterpret_cast<const void*>(cn4::conv3x3_n4_kernel), 100 * (512 + 8) * 2 + 4 * 9 * 512 * 2);
    if (ea != hipSuccess) return udt_set_hip_error(ea);
    UdtProfScope profn(0, s);
    if (profn.rec) {
      char tag[96];
      snprintf(tag, sizeof(tag), "conv_n4 M=%d N=%d K=%d %dx%d", d->M, d->N, d->K, d->Hin, d->Win);
      udt_prof_tag(profn.rec, tag);
    }
    hipLaunchKernelGGL(cn4::conv3x3_n4_kernel, dim3((unsigned)(cp.B * cp.tiles_x * cp.tiles_y)), dim3(256), smem, s, cp);
    UDT_CHECK_LAUNCH();
    return UDT_OK;
  }
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
  p.colscale = d->colscale;
  p.in_scsh = d->in_scsh;
  p.in_act = d->in_act;
  p.colstats = d->colstats;
  p.slabs = nullptr;
  if (d->colstats && colstats_rows(d) == 0) return UDT_ERR_BAD_ARG;     // ask udt_gemm_colstats_rows first
  if (d->in_scsh && !udt_gemm_in_scsh_ok(d)) return UDT_ERR_BAD_ARG;
  if (d->in_act != 0 && d->in_act != 1) return UDT_ERR_BAD_ARG;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.lda = d->lda; p.ldo = d->ldo; p.ldr = d->ldr;
  p.ldw = d->ldw > 0 ? d->ldw : d->K;
  if (p.ldw < d->K || p.ldw % 8 != 0) return UDT_ERR_BAD_SHAPE;
  if (mx8 && d->lda < d->K) return UDT_ERR_BAD_SHAPE;
  p.sA = d->stride_a; p.sW = d->stride_w; p.sO = d->stride_out; p.sR = d->stride_res;
  p.Hin = d->Hin; p.Win = d->Win; p.C1 = d->C1; p.C2 = d->C2; p.Hout = d->Hout; p.Wout = d->Wout;
  p.ksz = d->ksize; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l; p.ups = d->upsample ? 1 : 0;
  p.rows_per_batch = d->rows_per_batch > 0 ? d->rows_per_batch : d->M;
  p.ldrv = d->ld_rowvec > 0 ? d->ld_rowvec : d->N;
  if (d->flags & ~PUBLIC_FLAGS) return UDT_ERR_BAD_ARG;
  p.flags = d->flags | (g_rows_epi.load(std::memory_order_relaxed) ? 0 : INTERNAL_DIRECT_EPI);
#ifdef UDT_MEASURE
  p.flags |= g_dbg_bits.load(std::memory_order_relaxed);
#endif
  p.n_block = 1;
  p.alpha = d->alpha;
  const int cls = (conv && d->ksize == 3) ? 0 : 1;
  {
    LeanPlan lt;
    if (lean_plan(d, lt, d->colstats != nullptr)) {
      lg::LParams lp;
      lp.colstats = d->colstats;
      const bool conv1 = (d->flags & UDT_GEMM_CONV) != 0;
      lp.a = p.a; lp.a2 = (conv1 && d->C2 > 0) ? p.a2 : nullptr; lp.w = p.w; lp.bias = p.bias; lp.res = p.res; lp.rowvec = p.rowvec;
      lp.ln_s = d->ln_colsum;
      lp.out = reinterpret_cast<uint16_t*>(d->out);
      lp.M = d->M; lp.N = d->N; lp.K = d->K;
      lp.lda = conv1 ? d->C1 : p.lda; lp.ldw = p.ldw; lp.ldo = p.ldo; lp.ldr = p.ldr; lp.ldrv = p.ldrv; lp.rows_per_batch = p.rows_per_batch;
      lp.lda2 = conv1 ? d->C2 : 0;
      lp.kt_split = (conv1 && d->C2 > 0) ? d->C1 / BK : lt.nkt;
      lp.a2_bytes = (unsigned)((long long)d->M * (conv1 ? d->C2 : 0) * 2);
      lp.alpha = d->alpha; lp.ln_eps = d->ln_eps;
      lp.tiles_m = lt.tiles_m; lp.tiles_n = lt.tiles_n; lp.n_block = lt.n_block; lp.tiles = lt.tiles;
      lp.nkt = lt.nkt; lp.splitk = lt.splitk; lp.kt_per = lt.kt_per;
      lp.a_bytes = (unsigned)((long long)d->M * lp.lda * elem_bytes(d));
      lp.w_bytes = (unsigned)((long long)d->N * p.ldw * elem_bytes(d));
      lp.a_scale = mx8 ? reinterpret_cast<const uint32_t*>(d->a_scale) : nullptr;
      lp.colscale = d->colscale;
      lp.rowstat_in = d->rowstat_in; lp.rowstat_in_parts = d->rowstat_in_parts;
      lp.q8_out = reinterpret_cast<uint8_t*>(d->q8_out); lp.q8_scale = reinterpret_cast<uint32_t*>(d->q8_scale); lp.ld_q8 = d->ld_q8;
      lp.rowstat_out = d->rowstat_out;
      lp.q8_fixed_col = d->q8_fixed_col > 0 ? d->q8_fixed_col : 0x7fffffff;
      lp.q8_fixed_mul = d->q8_fixed_mul;
      lp.G = lt.G;
      lp.counters = nullptr; lp.slabs = nullptr;
      if (lt.splitk > 1) {
        if (!workspace || workspace_bytes < lean_workspace(lt)) return UDT_ERR_WORKSPACE;
        lp.counters = reinterpret_cast<int*>(workspace);
        lp.slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + G8_HEADER_BYTES);
      }
      UdtProfScope profl(cls, s);
      if (profl.rec) {
        char tag[96];
        snprintf(tag, sizeof(tag), "lean%d%s%s M=%d N=%d K=%d fl=0x%x ln=%d tile=%dx%d units=%d splitk=%d", lt.cfg, mx8 ? "-mx8" : "", d->q8_out ? "+q8" : "",
                 d->M, d->N, d->K, d->flags, d->ln_colsum ? 1 : 0, lt.bm, lt.bn, lt.tiles * lt.splitk, lt.splitk);
        udt_prof_tag(profl.rec, tag);
      }
      const bool geglu = (d->flags & UDT_GEMM_GEGLU) != 0, ln = d->ln_colsum != nullptr;
      hipError_t el;
      el = udt_lean_launch_gemm(lt.cfg, &lp, lt.smem, lt.G, geglu ? 1 : 0, ln ? 1 : 0, s);
      if (el != hipSuccess) return udt_set_hip_error(el);
      return UDT_OK;
    }
  }
  if (d->ln_colsum) return UDT_ERR_BAD_ARG;            // the LayerNorm-folded form exists on the lean kernels only
  {
    lg::C3Params c3;
    if (lean_conv_plan(d, c3, d->colstats != nullptr)) {
      c3.colstats = d->colstats;
      c3.a = p.a; c3.w = p.w; c3.bias = p.bias; c3.res = p.res; c3.rowvec = p.rowvec; c3.out = reinterpret_cast<uint16_t*>(d->out);
      if (c3.splitk > 1) {
        if (!workspace || workspace_bytes < lean_conv_workspace(c3)) return UDT_ERR_WORKSPACE;
        c3.counters = reinterpret_cast<int*>(workspace);
        c3.slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + G8_HEADER_BYTES);
      }
      UdtProfScope profc(cls, s);
      if (profc.rec) {
        char tag[96];
        snprintf(tag, sizeof(tag), "%s%s M=%d N=%d K=%d %dx%d tile=%dx%d units=%d splitk=%d nb=%d", c3.geo == 3 ? "wconv3" : "lconv3", d->upsample ? "+up" : "", d->M, d->N, d->K,
                 d->Hin, d->Win, c3.tw, c3.th, c3.tiles * c3.splitk, c3.splitk, c3.n_block);
        udt_prof_tag(profc.rec, tag);
      }
      const hipError_t ec = udt_lean_launch_conv3(&c3, s);
      if (ec != hipSuccess) return udt_set_hip_error(ec);
      return UDT_OK;
    }
  }
  {
    c3p::CParams cp;
    if (conv3p_geometry(d, cp.geo, d->in_scsh != nullptr)) {
      const TilePlan t3 = plan_tiles3p(d, cp.geo);
      p.tiles_m = t3.tiles_m; p.tiles_n = t3.tiles_n; p.tiles_per_batch = t3.tiles_m * t3.tiles_n;
      {
        // XCD-aware tile order.  An XCD's L2 sees ~S = min(tiles, workgroups) / 8 neighbouring tiles at a time; taking
        // them as (S / nb) patches x nb weight tiles costs (S / nb) * patch + nb * weight-tile bytes of L2 fills:
        // least at nb = sqrt(S * patch / weight tile)
        const double patch = (double)cp.geo.prows_img * cp.geo.NI * cp.geo.C * 2.0;
        const double wtile = 9.0 * t3.bn * cp.geo.C * 2.0;
        const double S = (double)(t3.tiles < t3.G ? t3.tiles : t3.G) / 8.0;
        int nb = (int)(std::sqrt((S < 1.0 ? 1.0 : S) * patch / wtile) + 0.5);
        if (nb < 1) nb = 1;
        if (nb > t3.tiles_n) nb = t3.tiles_n;
        p.n_block = nb;
      }
      p.n_ktiles = t3.nkt;
      p.total_iters = t3.total;
      p.iters_per_wg = t3.ipw;
      p.G = t3.G;
      const long long in_px = (long long)cp.geo.B * d->Hin * d->Win;       // (= M unless the launch upsamples)
      cp.base.a_bytes = (unsigned)(in_px * d->C1 * 2);
      cp.a2_bytes = (unsigned)((long long)d->M * d->C2 * 2);
      cp.scsh_bytes = (unsigned)((long long)cp.geo.B * cp.geo.chunks * 512);
      cp.base.w_bytes = (unsigned)((long long)d->N * p.ldw * 2);
      cp.base.g = p;
      cp.base.flags = nullptr; cp.base.err = nullptr; cp.base.slab_base = nullptr;
      if (t3.fixup) {
        const size_t need = G8_HEADER_BYTES + (size_t)t3.G * t3.bm * t3.bn * sizeof(float);
        if (!workspace || workspace_bytes < need) return UDT_ERR_WORKSPACE;
        cp.base.flags = reinterpret_cast<int*>(workspace);
        cp.base.err = cp.base.flags + 1023;
        cp.base.slab_base = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + G8_HEADER_BYTES);
      }
      UdtProfScope prof3(cls, s);
      if (prof3.rec) {
        char tag[96];
        snprintf(tag, sizeof(tag), "conv3p%s%s M=%d N=%d K=%d %dx%d tile=%dx%dx%d bn=%d G=%d ipw=%d", d->in_scsh ? "+gn" : "", d->upsample ? "+up" : "", d->M,
                 d->N, d->K, d->Hin, d->Win, cp.geo.TW, cp.geo.TH, cp.geo.NI, t3.bn, t3.G, t3.ipw);
        udt_prof_tag(prof3.rec, tag);
      }
      hipError_t e3;
      const bool st3 = d->colstats != nullptr;             // (GroupNorm-on-patch instances only: conv3p_geometry)
      if (t3.bn == 160) e3 = st3 ? launch3p<8, 1, 1, 5, true, true>(cp, t3, s) : launch3p<8, 1, 1, 5, true, false>(cp, t3, s);
      else e3 = st3 ? launch3p<4, 2, 2, 2, true, true>(cp, t3, s) : launch3p<4, 2, 2, 2, true, false>(cp, t3, s);
      if (e3 != hipSuccess) return udt_set_hip_error(e3);
      return UDT_OK;
    }
  }
  if (use_gemm8(d)) {
    const TilePlan t8 = plan_tiles8(d);
    p.tiles_m = t8.tiles_m; p.tiles_n = t8.tiles_n; p.tiles_per_batch = t8.tiles_m * t8.tiles_n;
    const int n_block_knob = g_n_block.load(std::memory_order_relaxed);
    // Tile order of plain GEMMs (decode_tile): N-tiles in blocks of n_block, N-fastest inside a block.  Automatic rule,
    // from the sweep over the UNet's shapes in profiles/r02_gemm_shapes_tile_order.txt (same-box, hipGraph-timed):
    //   few M-tiles (<= 8: the 16x16 / 8x8 levels) or a transposed output -> one block (N-fastest): the 256-row A tile
    //     stays in the XCD's L2 while its (half as large) weight tiles stream (2048x1280x5120: 64 -> 52 us);
    //   wide outputs (>= 8 N-tiles)                                      -> weight blocks of ~4 MiB;
    //   otherwise                                                          -> M-fastest (N = 640 at 64x64 measured slower blocked).
    if (n_block_knob != 0 && !conv) {
      const long long wt = (long long)t8.bn * d->K * elem_bytes(d);
      long long nb = 1;
      if (n_block_knob > 0) nb = n_block_knob;
      else if (t8.tiles_m <= 8 || trans) nb = t8.tiles_n;
      else if (t8.tiles_n >= 8) nb = (4LL << 20) / (wt > 0 ? wt : 1);
      if (nb < 1) nb = 1;
      if (nb > t8.tiles_n) nb = t8.tiles_n;
      p.n_block = (int)nb;
    }
    p.n_ktiles = t8.nkt;
    p.total_iters = t8.total;
    p.iters_per_wg = t8.ipw;
    p.G = t8.G;
    g8::Params pp;
    pp.g = p;
    pp.flags = nullptr; pp.err = nullptr; pp.slab_base = nullptr;
    pp.a_bytes = conv ? 0u : (unsigned)((long long)d->M * d->lda * elem_bytes(d));
    pp.w_bytes = (unsigned)((long long)d->N * p.ldw * elem_bytes(d));
    if (t8.fixup) {
      const size_t need = G8_HEADER_BYTES + (size_t)t8.G * t8.bm * t8.bn * sizeof(float);
      if (!workspace || workspace_bytes < need) return UDT_ERR_WORKSPACE;
      pp.flags = reinterpret_cast<int*>(workspace);
      pp.err = pp.flags + 1023;
      pp.slab_base = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + G8_HEADER_BYTES);
    }
    UdtProfScope prof8(cls, s);
    if (prof8.rec) {
      char tag[96];
      snprintf(tag, sizeof(tag), "gemm8 M=%d N=%d K=%d conv=%d ks=%d fl=0x%x tile=%dx%d G=%d ipw=%d b=%d",
               d->M, d->N, d->K, conv ? 1 : 0, d->ksize, d->flags, t8.bm, t8.bn, t8.G, t8.ipw, batch);
      udt_prof_tag(prof8.rec, tag);
    }
    hipError_t e8;
    const bool st = d->colstats != nullptr;         // statistics-emitting epilogues are separate kernels
    if (t8.bn == 160) {
      if (st) e8 = conv ? launch8<8, 1, 1, 5, true, false, true>(pp, t8, s) : launch8<8, 1, 1, 5, false, false, true>(pp, t8, s);
      else e8 = conv ? launch8<8, 1, 1, 5, true, false>(pp, t8, s) : launch8<8, 1, 1, 5, false, false>(pp, t8, s);
    } else if (trans) {
      e8 = launch8<4, 2, 2, 2, false, true>(pp, t8, s);
    } else if (st) {
      e8 = conv ? launch8<4, 2, 2, 2, true, false, true>(pp, t8, s) : launch8<4, 2, 2, 2, false, false, true>(pp, t8, s);
    } else {
      e8 = conv ? launch8<4, 2, 2, 2, true, false>(pp, t8, s) : launch8<4, 2, 2, 2, false, false>(pp, t8, s);
    }
    if (e8 != hipSuccess) return udt_set_hip_error(e8);
    return UDT_OK;
  }

  p.tiles_m = t.tiles_m; p.tiles_n = t.tiles_n; p.tiles_per_batch = t.tiles_m * t.tiles_n;
  p.n_ktiles = t.nkt;
  p.total_iters = t.total;
  p.iters_per_wg = t.ipw;
  p.G = t.G;

  if (t.fixup) {
    const size_t need = G8_HEADER_BYTES + (size_t)t.G * 2 * SLAB_FLOATS * sizeof(float);
    if (!workspace || workspace_bytes < need) return UDT_ERR_WORKSPACE;
    p.slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + G8_HEADER_BYTES);
  }

  UdtProfScope prof(cls, s);
  if (prof.rec) {
    char tag[96];
    snprintf(tag, sizeof(tag), "gemm M=%d N=%d K=%d conv=%d ks=%d fl=0x%x tile=%dx%d G=%d ipw=%d b=%d", d->M, d->N, d->K,
             conv ? 1 : 0, d->ksize, d->flags, t.bm, t.bn, t.G, t.ipw, batch);
    udt_prof_tag(prof.rec, tag);
  }

  // the first-generation 4-wave kernel survives in ONE geometry, 256 x 64 tiles for outputs of <= 64 columns (the UNet's
  // 4-channel output convolution); its 128 x 128 instances (round 1) are gone: transposed / GEGLU outputs that narrow, or operands
  // beyond the 31-bit buffer offsets of the newer kernels, are not part of this path
  if (t.bm != 256) return UDT_ERR_BAD_SHAPE;
  const hipError_t e = conv ? launch_cfg<256, 64, 4, 1, true, false>(p, t, s) : launch_cfg<256, 64, 4, 1, false, false>(p, t, s);
  if (e != hipSuccess) return udt_set_hip_error(e);
  return UDT_OK;
}
