// conv3p.h — 3x3 / stride-1 / pad-1 convolution with LDS-staged input patches (included by gemm.hip).
//
// Why: the implicit-GEMM kernels are bound by the global->LDS fill rate (PMC: MFMA busy 46 %, zero bank conflicts,
// ~42 GB/s of LDS-DMA per CU), and a 3x3 convolution streams the SAME activation rows through LDS nine times, once per
// tap.  Here a workgroup owns a spatial block of 256 output pixels (32x8, 16x16 or four 8x8 images) x BN output
// channels and, per 64-channel chunk, stages the input patch INCLUDING ITS HALO once (<= 400 pixel rows x 128 B);
// the nine taps are nine K-tiles that read their A fragments from that patch at shifted row offsets, so only the
// weight tile (BN x 128 B) is fetched per K-tile: ~2.3x less LDS fill per MFMA than the gather formulation.
//
// Everything else is the 8-wave machinery of gemm8.h: 3-stage weight ring + double-buffered patch, counted vmcnt
// across raw barriers, lead/lag wave roles, persistent stream-K ranges over (tile, chunk, tap) with in-kernel finishing.
// K order inside a tile is (chunk, tap); the packed weight layout k = tap*C + c is unchanged.
#pragma once
#include <type_traits>

#ifndef UDT_C3P_PIPE
#define UDT_C3P_PIPE 1
#endif

namespace c3p {

using g8::NSTAGE;
using g8::NTHREADS;
using g8::Params;
using g8::raw_barrier;

// max pixel rows of a staged patch: 400 (4 x 10 x 10, the four-image 8x8 tile) with 128-channel weight tiles; the
// 160-channel configuration (N = 320) has 20 KiB weight stages, so with GroupNorm on the patch its patches are capped
// at 344 rows (34 x 10 and 18 x 18 tiles) to leave room for the scale/shift stage
constexpr int patch_rows(int TN, bool GN) { return (TN == 5 && GN) ? 344 : 400; }
constexpr int SCSH_BYTES = 2 * 4 * 512;            // [2 chunks in flight][<= 4 images][64 scale | 64 shift] fp32

struct Geo {            // spatial tiling of the output (= input) map
  int TW, TH, NI;       // tile width / height in pixels, images per tile (TW*TH*NI == 256)
  int tiles_x, tiles_y; // tiles per image row / column
  int img_groups;       // ceil(B / NI)
  int B, H, W, C;       // C = C1 + C2: channels of the (virtually concatenated) input
  int C1, C2;           // channels of source 1 / source 2 (0: one source); both multiples of 64
  int prow_w;           // TW + 2
  int prows_img;        // (TH + 2) * (TW + 2)
  int n_pieces;         // ceil(NI * prows_img / 8)
  int chunks;           // C / 64
};

struct CParams {
  Params base;              // base.a_bytes / base.w_bytes: buffer-descriptor extents (source 1, weights)
  Geo geo;
  unsigned a2_bytes;        // extent of source 2
  unsigned scsh_bytes;      // extent of the GroupNorm scale/shift table [B][chunks][2][64] fp32
};

UDT_DEVINL void wait_vmcnt_dyn(int n) {          // n is wave-uniform
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// epilogue with an explicit output-row table: mrow[tm] = global NHWC pixel index of this lane's row, or -1
template <int TM, int TN>
UDT_DEVINL void epilogue_rows_plain(const GemmParams& p, f32x16 (&acc)[TM][TN], const long long (&mrow)[TM], int n0, int col0,
                              int lane) {
  const int hi = lane >> 5;
  const int flags = p.flags;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const long long m = mrow[tm];
    if (m < 0) continue;
    const int b = (p.rowvec != nullptr) ? (int)(m / p.rows_per_batch) : 0;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + col0 + tn * 32 + q * 8 + hi * 4;
        if (n < p.N) {
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
          if (p.res) {
            const u32x2 rr = *reinterpret_cast<const u32x2*>(p.res + m * p.ldr + n);
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
            f32x4 ov = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + m * p.ldo + n) = ov;
          } else {
            u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(p.out) + m * p.ldo + n) = pk;
          }
        }
      }
  }
}

// the same with column statistics of the stored values (tn-outer so that one 32-column tile's sums are live at a time).
// `stats`: this wave block's row of the column statistics, fp32 [N][2], offset to the wave's first column
template <int TM, int TN>
UDT_DEVINL void epilogue_rows_stats(const GemmParams& p, f32x16 (&acc)[TM][TN], const long long (&mrow)[TM], int n0, int col0,
                              int lane, float* stats) {
  const int hi = lane >> 5;
  const int flags = p.flags;
  int bidx[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) bidx[tm] = (p.rowvec != nullptr && mrow[tm] >= 0) ? (int)(mrow[tm] / p.rows_per_batch) : 0;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    float cs[16], cq[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) cs[j] = cq[j] = 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const long long m = mrow[tm];
      if (m < 0) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + col0 + tn * 32 + q * 8 + hi * 4;
        if (n < p.N) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[tm][tn][q * 4 + r] * p.alpha;
          if (p.bias) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += bv[r];
          }
          if (p.rowvec) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(p.rowvec + (long long)bidx[tm] * p.ldrv + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += rv[r];
          }
          if (p.res) {
            const u32x2 rr = *reinterpret_cast<const u32x2*>(p.res + m * p.ldr + n);
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
            f32x4 ov = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + m * p.ldo + n) = ov;
          } else {
            u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(p.out) + m * p.ldo + n) = pk;
            if (stats) {                 // statistics of the values as stored (bf16-rounded), like the row epilogues
#pragma unroll
              for (int r = 0; r < 2; ++r) {
                const float lo = bf16_lo(pk[r]), hh = bf16_hi(pk[r]);
                cs[q * 4 + 2 * r] += lo; cq[q * 4 + 2 * r] += lo * lo;
                cs[q * 4 + 2 * r + 1] += hh; cq[q * 4 + 2 * r + 1] += hh * hh;
              }
            }
          }
        }
      }
    }
    if (stats) g8::colstat_emit_acc(cs, cq, lane, stats + tn * 64, p.N - (n0 + col0 + tn * 32));
  }
}

using g8::wait_vm;
using g8::buf_lds16;
using g8::OOB;
// WGM x WGN waves, each TM x TN MFMA tiles of 32x32; BM = 256 output pixels.
// The stream-K iteration unit is one 64-channel CHUNK (= 9 K-tiles, one per tap): ranges never cut a chunk, so the tap
// loop is fully unrolled — stage indices, tap offsets and every s_waitcnt immediate are compile-time constants, and the
// per-K-tile address work is one scalar offset per load (buffer descriptors: uniform base + per-lane voffset fixed for
// the whole tile + scalar soffset per K-tile; out-of-image halo pixels use an out-of-range voffset and read zeros).
//
// The input may be the channel concatenation of TWO NHWC sources (the UNet's skip concat, openaimodel.py:620): a chunk
// lies in exactly one of them (C1 % 64 == 0), so only the descriptor and the pixel stride change per chunk.
//
// GN = true: GroupNorm (+ SiLU) of the input is applied ON THE STAGED PATCH (reference chain GroupNorm32 -> SiLU ->
// conv, openaimodel.py:183-187,218-231): the per-(sample, channel) scale / shift of the chunk (udt_gn_finalize, from the
// producers' epilogue statistics) arrives by one more LDS-DMA next to the patch, and every wave rewrites the pieces IT
// staged — in-image rows only, the zero padding stays zero — as y = act(x * scale + shift) in fp32, rounded to bf16
// once.  The patch of chunk c+1 lands during taps 0..2 of chunk c; its pieces are transformed one per tap behind the
// MFMAs of taps 3..8 (VALU beside the other wave's matrix work), so only the first chunk of a segment pays for it.
// STATS = true: the epilogue also emits the output's per-(row slot, column) partial sums (udt_gemm_desc.colstats).
// The four (GN, STATS) combinations are separate kernels: the plain one is the round-1 kernel instruction for
// instruction (this kernel's register allocation is tight: extra code anywhere in it costs 5-10 % of the launch).
// UPS = true: nearest x2 upsampling folded into the convolution (reference Upsample.forward, openaimodel.py:99-101 /
// model.py:64-68: F.interpolate(scale 2, nearest) then conv3x3 pad 1).  The tile is 256 pixels of the UPSAMPLED map; the
// staged patch is the low-resolution region under it with its halo ((TW/2 + 2) x (TH/2 + 2) rows), and the tap (dy, dx)
// of output pixel (py, px) reads patch row ((py + dy - 1) >> 1) + 1, column ((px + dx - 1) >> 1) + 1.  ge.H / ge.W are the
// INPUT dimensions.  One source, no GroupNorm, no statistics (a separate kernel instance: the others are unchanged).
template <int WGM, int WGN, int TM, int TN, bool GN, bool STATS, bool UPS = false>
__global__ void __launch_bounds__(NTHREADS) conv3p_kernel(const CParams cp) {
  static_assert(!UPS || (!GN && !STATS && UDT_C3P_PIPE), "the upsampling variant covers the plain (pipelined) convolution");
  static_assert(WGM * WGN == 8 && WGM * TM * 32 == 256, "8 waves, 256 output pixels");
  constexpr int BN = WGN * TN * 32;
  constexpr int W_BYTES = BN * ROW_BYTES;
  constexpr int W_PIECES = BN / 8;                  // 16 or 20
  constexpr int NWP = (W_PIECES + 7) / 8;           // weight pieces per wave and K-tile (2 or 3, padded with duplicates)
  constexpr int PATCH_BYTES = patch_rows(TN, GN) * ROW_BYTES;
  constexpr int PP = (patch_rows(TN, GN) / 8 + 7) / 8;  // patch pieces per wave and chunk (6 or 7, padded with duplicates)
  constexpr int PPL = PP + (GN ? 1 : 0);            // + the scale/shift piece
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wring = smem;                              // NSTAGE weight stages
  char* const patches = smem + NSTAGE * W_BYTES;         // 2 patch stages
  char* const scsh_lds = patches + 2 * PATCH_BYTES;      // GN: [2][4 images][64 scale | 64 shift] fp32

  const GemmParams& p = cp.base.g;
  const Geo& ge = cp.geo;
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
  const int swz_w = (l31 >> 1) & 7;
  const int w_frag_row = (col0 + l31) * ROW_BYTES;
  const int chunks = p.n_ktiles;                         // iteration unit: chunk
  const int nch1 = ge.C1 >> 6;                           // chunks of source 1

  const int g = range_index(blockIdx.x, p.G);
  long long it = (long long)g * p.iters_per_wg;
  long long it_end = it + p.iters_per_wg;
  if (it_end > p.total_iters) it_end = p.total_iters;
  if (it >= it_end) return;

  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w), 0, cp.base.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.a), 0, cp.base.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_a2 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.a2 ? p.a2 : p.a), 0, p.a2 ? cp.a2_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_s =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GN ? p.in_scsh : nullptr), 0, GN ? cp.scsh_bytes : 0u, 0x00020000);

  // ---- per-lane state of the current tile ------------------------------------------------------------------
  // piece index of load i of this wave: wave + 8*i, wrapped back by multiples of 8 when past the last piece (the
  // duplicate re-writes identical bytes; it keeps the number of loads per wave uniform = compile-time wait counts)
  int w_piece[NWP];
  unsigned w_voff[NWP];
#pragma unroll
  for (int i = 0; i < NWP; ++i) {
    int idx = wave + 8 * i;
    while (idx >= W_PIECES) idx -= 8;
    w_piece[i] = idx;
  }
  int p_piece[PP];
  unsigned p_voff[PP];                 // !GN: byte offset of this lane's 16 bytes of the patch row in the (single) source
  int p_info[PP];                      // GN: input pixel index of this lane's patch row | image-in-tile << 28, or -1 (padding)
#pragma unroll
  for (int i = 0; i < PP; ++i) {
    int idx = wave + 8 * i;
    while (idx >= ge.n_pieces) idx -= 8;
    p_piece[i] = idx;
  }
  // 16-byte slot of this lane inside every patch row it stages: row = piece * 8 + l3 and piece = wave (mod 8), so the
  // XOR swizzle ((row >> 1) & 7) depends on the lane and the wave's parity only
  const int koff = (pslot ^ ((((wave & 1) << 2) + (l3 >> 1)) & 7)) * 8;       // first of the lane's 8 channels in a chunk
  unsigned s_voff = OOB;               // GN: this lane's 16 bytes of the tile's scale/shift rows
  int a_prow[TM];                      // patch row of this lane's output pixel at tap (0,0)
  long long mrow[TM];                  // global output pixel index of this lane's rows (epilogue)

  auto prepare = [&](int tile_m, int n0) {
    const int per_img = ge.tiles_x * ge.tiles_y;
    const int ig = tile_m / per_img;
    const int r = tile_m - ig * per_img;
    const int ty = r / ge.tiles_x;
    const int tx = r - ty * ge.tiles_x;
    const int y0 = ty * ge.TH, x0 = tx * ge.TW, b0 = ig * ge.NI;
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
      const int row = w_piece[i] * 8 + l3;
      const int kw = (pslot ^ ((row >> 1) & 7)) * 8;
      const int n = n0 + row;
      w_voff[i] = (n < p.N) ? (unsigned)(((long long)n * p.ldw + kw) * 2) : OOB;
    }
#pragma unroll
    for (int i = 0; i < PP; ++i) {
      const int prow = p_piece[i] * 8 + l3;
      const int img = prow / ge.prows_img;
      const int rem = prow - img * ge.prows_img;
      const int yy = rem / ge.prow_w;
      const int xx = rem - yy * ge.prow_w;
      const int gy = (UPS ? (y0 >> 1) : y0) + yy - 1, gx = (UPS ? (x0 >> 1) : x0) + xx - 1, b = b0 + img;
      const bool ok = (img < ge.NI) && (b < ge.B) && ((unsigned)gy < (unsigned)ge.H) && ((unsigned)gx < (unsigned)ge.W);
      if constexpr (GN) p_info[i] = ok ? ((((b * ge.H + gy) * ge.W + gx)) | (img << 28)) : -1;
      else p_voff[i] = ok ? (unsigned)(((((long long)b * ge.H + gy) * ge.W + gx) * ge.C + koff) * 2) : OOB;
    }
    if constexpr (GN) {
      // scale/shift stage of a chunk: [image][64 scale | 64 shift] = 512 B per image; piece (wave & 1) of 1 KiB
      const int img = ((wave & 1) << 1) + (lane >> 5);
      const int b = b0 + img;
      s_voff = (img < ge.NI && b < ge.B) ? (unsigned)(((long long)b * chunks * 512) + (lane & 31) * 16) : OOB;
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int ml = row0 + tm * 32 + l31;
      const int per = ge.TW * ge.TH;
      const int img = ml / per;
      const int rem = ml - img * per;
      const int py = rem / ge.TW;
      const int px = rem - py * ge.TW;
      a_prow[tm] = UPS ? (py | (px << 16)) : (img * ge.prows_img + py * ge.prow_w + px);
      const int b = b0 + img;
      if constexpr (UPS) mrow[tm] = (b < ge.B) ? (((long long)b * (2 * ge.H) + (y0 + py)) * (2 * ge.W) + (x0 + px)) : -1;
      else mrow[tm] = (b < ge.B) ? (((long long)b * ge.H + (y0 + py)) * ge.W + (x0 + px)) : -1;
    }
  };

  auto issue_w = [&](int st, int c, int tap) {
    const int soff = (tap * ge.C + c * 64) * 2;
    char* wbuf = wring + st * W_BYTES;
#pragma unroll
    for (int i = 0; i < NWP; ++i) buf_lds16(rsrc_w, wbuf + w_piece[i] * 1024, w_voff[i], soff);
  };
  auto issue_patch = [&](int c) {
    char* pbuf = patches + (c & 1) * PATCH_BYTES;
    if constexpr (GN) {
      const bool second = c >= nch1;                   // wave-uniform: the chunk lies in source 2
      const int cs = second ? ge.C2 : ge.C1;
      const int soff = (second ? c - nch1 : c) * 128;
#pragma unroll
      for (int i = 0; i < PP; ++i) {
        const unsigned voff = (p_info[i] >= 0) ? (unsigned)(((long long)(p_info[i] & 0x0fffffff) * cs + koff) * 2) : OOB;
        buf_lds16(second ? rsrc_a2 : rsrc_a, pbuf + p_piece[i] * 1024, voff, soff);
      }
      buf_lds16(rsrc_s, scsh_lds + (c & 1) * 2048 + (wave & 1) * 1024, s_voff, c * 512);
    } else {
      const int soff = c * 128;
#pragma unroll
      for (int i = 0; i < PP; ++i) buf_lds16(rsrc_a, pbuf + p_piece[i] * 1024, p_voff[i], soff);
    }
  };
  // GN: rewrite piece i of chunk c's patch (staged by THIS wave, already landed) as act(x * scale + shift).  A lane's 8
  // channels are the same for every piece (koff), so with one image per tile the chunk's 16 scale / shift values are
  // read from LDS once (load_scsh) and kept in registers for all pieces; the four-image 8x8 tile reads them per piece.
  f32x4 gs0 = {0.f, 0.f, 0.f, 0.f}, gs1 = gs0, gh0 = gs0, gh1 = gs0;
  const bool one_image = (ge.NI == 1);
  auto load_scsh = [&](int c, int img) {
    const float* sc = reinterpret_cast<const float*>(scsh_lds + (c & 1) * 2048 + img * 512) + koff;
    gs0 = *reinterpret_cast<const f32x4*>(sc);
    gs1 = *reinterpret_cast<const f32x4*>(sc + 4);
    gh0 = *reinterpret_cast<const f32x4*>(sc + 64);
    gh1 = *reinterpret_cast<const f32x4*>(sc + 68);
  };
  auto transform_piece = [&](auto i_c, int c) {
    constexpr int I = decltype(i_c)::value;
    if constexpr (GN && I < PP) {
      if (wave + 8 * I < ge.n_pieces) {              // not a duplicate (wave-uniform)
        const int info = p_info[I];
        char* cell = patches + (c & 1) * PATCH_BYTES + p_piece[I] * 1024 + lane * 16;
        if (!one_image) load_scsh(c, (info >> 28) & 3);
        if (info >= 0) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(cell);
          float o[8];
          o[0] = bf16_lo(v[0]) * gs0[0] + gh0[0];
          o[1] = bf16_hi(v[0]) * gs0[1] + gh0[1];
          o[2] = bf16_lo(v[1]) * gs0[2] + gh0[2];
          o[3] = bf16_hi(v[1]) * gs0[3] + gh0[3];
          o[4] = bf16_lo(v[2]) * gs1[0] + gh1[0];
          o[5] = bf16_hi(v[2]) * gs1[1] + gh1[1];
          o[6] = bf16_lo(v[3]) * gs1[2] + gh1[2];
          o[7] = bf16_hi(v[3]) * gs1[3] + gh1[3];
          if (p.in_act == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = silu_f(o[j]);
          }
          u32x4 ov = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
          *reinterpret_cast<u32x4*>(cell) = ov;
        }
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>;
  using I5 = std::integral_constant<int, 5>;
  using I6 = std::integral_constant<int, 6>;
  auto lds_writes_done = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

  int tile = (int)(it / chunks);
  int c0 = (int)(it - (long long)tile * chunks);
  // tile order: N-tiles in blocks of p.n_block (decode_tile in gemm.hip; 1 = M-fastest) — with each XCD walking one
  // contiguous slice of the tile space, its L2 then serves (slice / n_block) patches to n_block weight tiles each
  int tile_n, tile_m, tile_b;
  decode_tile<1, 1>(p, tile, tile_b, tile_m, tile_n);
  int n0 = tile_n * BN;
  prepare(tile_m, n0);
  issue_patch(c0);
  issue_w(0, c0, 0);
  issue_w(1, c0, 1);

  while (true) {
    int c1 = chunks;
    if ((long long)(c1 - c0) > it_end - it) c1 = c0 + (int)(it_end - it);

    if constexpr (GN) {
      // first chunk of the segment: its patch (+ scale/shift) was requested ahead of the two weight tiles in flight
      wait_vm<2 * NWP>();
      if (one_image) load_scsh(c0, 0);
      transform_piece(I0{}, c0); transform_piece(I1{}, c0); transform_piece(I2{}, c0); transform_piece(I3{}, c0);
      transform_piece(I4{}, c0); transform_piece(I5{}, c0); transform_piece(I6{}, c0);
      lds_writes_done();
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int k = 0; k < TN; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;

    bf16x8_t fx[4][TM], fw[4][TN];
    auto mfma_all = [&]() {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32(fw[ks][tn], fx[ks][tm], acc[tm][tn]);
    };

    // One K-tile step.  DX (= tap % 3 = weight-ring stage) is compile time, dy (tap / 3) is a run-time loop index:
    // wait for W(c,tap) [+ the patch at tap 0], barrier, keep two K-tiles of weights in flight, request the next
    // chunk's patch at tap 0.
    auto head = [&](auto dx_c, int c, int dy, bool nxt) {
      constexpr int DX = decltype(dx_c)::value;
      const int tap = dy * 3 + DX;
      if (DX == 2 && dy == 2) {                                   // tap 8: next weights belong to the next chunk
        if (nxt) wait_vm<NWP>(); else wait_vm<0>();
      } else if (DX != 0 && dy == 0 && nxt) {                     // taps 1, 2: the patch requested at tap 0 may be in flight
        wait_vm<NWP + PPL>();
      } else {
        wait_vm<NWP>();
      }
      raw_barrier();
      if (tap <= 6) issue_w((DX + 2) % 3, c, tap + 2);
      else if (nxt) issue_w((DX + 2) % 3, c + 1, tap + 2 - 9);
      if (DX == 0 && dy == 0 && nxt) issue_patch(c + 1);
    };
    auto read_all = [&](auto dx_c, int c, int dy) {
      constexpr int DX = decltype(dx_c)::value;
      const char* pbuf = patches + (c & 1) * PATCH_BYTES;
      const char* wbuf = wring + DX * W_BYTES;
      int arow[TM], aswz[TM];
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int prow = a_prow[t] + dy * ge.prow_w + DX;
        arow[t] = prow * ROW_BYTES;
        aswz[t] = (prow >> 1) & 7;
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int t = 0; t < TM; ++t) fx[ks][t] = lds_read_frag(pbuf + arow[t] + (((ks * 2 + hi) ^ aswz[t]) << 4));
        const int slot = ((ks * 2 + hi) ^ swz_w) << 4;
#pragma unroll
        for (int t = 0; t < TN; ++t) fw[ks][t] = lds_read_frag(wbuf + w_frag_row + t * 32 * ROW_BYTES + slot);
      }
    };
    // GN: the next chunk's patch has landed once the tap-3 wait has passed (it was requested before W(c, 3)); this
    // wave's pieces are rewritten one per tap behind the MFMAs of taps 3..8
    auto transform_next = [&](auto dx_c, int c, int dy, bool nxt) {
      constexpr int DX = decltype(dx_c)::value;
      if constexpr (GN) {
        if (nxt && dy >= 1) {
          if (dy == 1) {
            if (DX == 0 && one_image) load_scsh(c + 1, 0);
            transform_piece(std::integral_constant<int, DX>{}, c + 1);
          } else {
            transform_piece(std::integral_constant<int, 3 + DX>{}, c + 1);
            if (DX == 2) {
              transform_piece(I6{}, c + 1);
              lds_writes_done();          // LDS ops retire in order: one wait ahead of the next chunk's first barrier
            }
          }
        }
      }
    };
    // (the lead/lag wave-role split of gemm8.h measured no gain once the per-K-tile instruction overhead was gone, and
    //  holding a K-tile of operands across the barrier cost spills here — all waves run the same straight loop)
#if UDT_C3P_PIPE
    if constexpr (!GN) {
      // Software-pipelined taps: tap t+1 reads its A fragments from the SAME staged patch as tap t (tap 8 -> the next
      // chunk's patch, complete since the tap-3 barrier), so they are fetched into a second fragment set while tap t's
      // MFMAs run; only the weight fragments are read behind the tap's barrier.  (profiles/r02_conv3p_loop_decomposition:
      // the fragment reads otherwise add 0.29 us to a 0.46 us K-tile instead of hiding under its MFMAs.)
      bf16x8_t fy[4][TM];
      auto read_a = [&](auto dx_c, int c, int dy, bf16x8_t (&dst)[4][TM]) {
        constexpr int DX = decltype(dx_c)::value;
        const char* pbuf = patches + (c & 1) * PATCH_BYTES;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          int prow;
          if constexpr (UPS) {
            const int py = a_prow[t] & 0xffff, px = a_prow[t] >> 16;
            prow = (((py + dy - 1) >> 1) + 1) * ge.prow_w + (((px + DX - 1) >> 1) + 1);
          } else {
            prow = a_prow[t] + dy * ge.prow_w + DX;
          }
          const int arow = prow * ROW_BYTES, aswz = (prow >> 1) & 7;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) dst[ks][t] = lds_read_frag(pbuf + arow + (((ks * 2 + hi) ^ aswz) << 4));
        }
      };
      auto read_w = [&](auto dx_c) {
        constexpr int DX = decltype(dx_c)::value;
        const char* wbuf = wring + DX * W_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int slot = ((ks * 2 + hi) ^ swz_w) << 4;
#pragma unroll
          for (int t = 0; t < TN; ++t) fw[ks][t] = lds_read_frag(wbuf + w_frag_row + t * 32 * ROW_BYTES + slot);
        }
      };
      auto mfma_from = [&](bf16x8_t (&src)[4][TM]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32(fw[ks][tn], src[ks][tm], acc[tm][tn]);
        // order: the tap's weight fragments, then its MFMAs with the next tap's A reads spread between them
        __builtin_amdgcn_sched_group_barrier(0x100, 4 * TN, 0);
#pragma unroll
        for (int i = 0; i < 4 * TM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, TN, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      };
      for (int c = c0; c < c1; ++c) {
        const bool nxt = (c + 1 < c1);
        for (int dy = 0; dy < 3; ++dy) {
          head(I0{}, c, dy, nxt);
          if (c == c0 && dy == 0) read_a(I0{}, c, 0, fx);          // first tap of the segment: nothing was prefetched
          read_w(I0{}); read_a(I1{}, c, dy, fy); mfma_from(fx);
          head(I1{}, c, dy, nxt);
          read_w(I1{}); read_a(I2{}, c, dy, fx); mfma_from(fy);
          head(I2{}, c, dy, nxt);
          // next tap: (c, dy + 1, 0), or tap 0 of the next chunk (a stale but in-range read after the last chunk)
          read_w(I2{}); read_a(I0{}, c + (dy == 2 ? 1 : 0), dy == 2 ? 0 : dy + 1, fy); mfma_from(fx);
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int t = 0; t < TM; ++t) fx[ks][t] = fy[ks][t];
        }
      }
    } else
#endif
    for (int c = c0; c < c1; ++c) {
      const bool nxt = (c + 1 < c1);
      for (int dy = 0; dy < 3; ++dy) {
        head(I0{}, c, dy, nxt); read_all(I0{}, c, dy); mfma_all(); transform_next(I0{}, c, dy, nxt);
        head(I1{}, c, dy, nxt); read_all(I1{}, c, dy); mfma_all(); transform_next(I1{}, c, dy, nxt);
        head(I2{}, c, dy, nxt); read_all(I2{}, c, dy); mfma_all(); transform_next(I2{}, c, dy, nxt);
      }
    }

    const int j0 = c0, j1 = c1;       // (names used by the finishing code below: units are chunks)
    const bool noxchg = UDT_DBG(p.flags, 28);       // measurement modes (udt_debug_set): wrong results
    const bool full = noxchg || ((j0 == 0) && (j1 == p.n_ktiles));
    const bool publish = !noxchg && (j0 > 0);
    const int cur_tile = tile, cur_n0 = n0, cur_tile_m = tile_m;
    long long cur_mrow[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) cur_mrow[t] = mrow[t];
    it += j1 - j0;
    const bool more = it < it_end;
    if (more) {
      raw_barrier();
      tile = (int)(it / chunks);
      c0 = (int)(it - (long long)tile * chunks);
      decode_tile<1, 1>(p, tile, tile_b, tile_m, tile_n);
      n0 = tile_n * BN;
      prepare(tile_m, n0);
      issue_patch(c0);
      issue_w(0, c0, 0);
      issue_w(1, c0, 1);
    }

    if (publish) {
      f32x4* slab = reinterpret_cast<f32x4*>(cp.base.slab_base + (long long)g * (256 * BN));
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
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        // slab stores were write-through (sc1) and are drained: no L2 write-back fence needed
        __hip_atomic_store(cp.base.flags + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (!UDT_DBG(p.flags, 27)) {
      if (!full) {
        const long long tile_end = ((long long)cur_tile + 1) * p.n_ktiles;
        const int g_last = (int)((tile_end - 1) / p.iters_per_wg);
        int* const bcast = reinterpret_cast<int*>(wring + 2 * W_BYTES);      // weight-ring stage 2 is idle here
        if (!more) __syncthreads();        // (with `more` the barrier ahead of the prefetch already closed the ring)
        if (tid == 0) {
          int ok = 1;
          for (int pg = g + 1; pg <= g_last && ok; ++pg) {
            int spins = 0;
            while (__hip_atomic_load(cp.base.flags + pg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
              __builtin_amdgcn_s_sleep(8);
              if (++spins > g8::SPIN_LIMIT) {      // partner not resident: flag the workspace, poison the tile (gemm8.h)
                __hip_atomic_store(cp.base.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
            const f32x4* slab = reinterpret_cast<const f32x4*>(cp.base.slab_base + (long long)pg * (256 * BN));
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
            __hip_atomic_store(cp.base.flags + pg, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // (the row-coalesced LDS epilogue of gemm8.h was measured here too: 109 vs 95 us per launch — the cross-lane
      //  row table and the extra live state cost more than the whole-line stores return; direct stores stay)
      // column statistics: one slot per wave row block (TM * 32 pixels of one image), slots in tile order
      if constexpr (STATS) {
        float* stats = p.colstats + (((long long)cur_tile_m * WGM + wm) * p.N + cur_n0 + col0) * 2;
        epilogue_rows_stats<TM, TN>(p, acc, cur_mrow, cur_n0, col0, lane, stats);
      } else {
        epilogue_rows_plain<TM, TN>(p, acc, cur_mrow, cur_n0, col0, lane);
      }
    }
    if (!more) break;
  }
}

}  // namespace c3p
