// lean_params.h — kernel argument blocks of the lean / wide families (lean.h, wide.h): filled by the dispatcher in gemm.hip,
// consumed by the kernels in lean.hip (plain data, no device code).
#pragma once

namespace lg {

struct LParams {
  const uint16_t* a;
  const uint16_t* a2;      // second A source (channel concat along K: the UNet's 1x1 skip convolutions), or nullptr
  const uint16_t* w;
  const float* bias;       // [N] fp32 or nullptr (LN: c_n)
  const uint16_t* res;     // bf16 [M, ldr] or nullptr
  const float* rowvec;     // fp32 [M / rows_per_batch, ldrv] or nullptr
  const float* ln_s;       // LN: s_n = sum_k W'_nk, fp32 [N]
  uint16_t* out;
  int M, N, K;
  int lda, ldw, ldo, ldr, ldrv, rows_per_batch;
  float alpha, ln_eps;
  int tiles_m, tiles_n, n_block, tiles;
  int nkt;                 // K tiles of the problem
  int splitk, kt_per;      // K slices per tile, K tiles per slice
  unsigned a_bytes, w_bytes;
  int kt_split, lda2;      // K tiles [0, kt_split) come from `a`, the rest from `a2` (row stride lda2); kt_split = nkt: one source
  unsigned a2_bytes;
  int G;                   // launched workgroups (a multiple of 8 when > 8)
  int* counters;           // split-K: [tiles] arrival tickets, zero between launches
  float* slabs;            // split-K: [tiles * splitk][BM * BN] fp32
  float* colstats;         // STATS kernels: fp32 [row slots][N][2] (sum, sum of squares) of the stored values, one slot per wave row block
  // ---- MX8 (UDT_GEMM_MX8: `a` / `w` hold e4m3 bytes; lda / ldw / K count elements = bytes; a K tile is 128 elements) ----
  const uint32_t* a_scale; // FP8 kernels: E8M0 block scales of A, uint32 [K / 128][M] (common.h "MX8 activations")
  const float* colscale;   // FP8 kernels: fp32 [N] per-output-channel weight scales (multiply the accumulators)
  const float* rowstat_in; // FP8 + LN kernels: fp32 [rowstat_in_parts][M][2] partial (sum, sum of squares) of the rows of A
  int rowstat_in_parts;
  uint8_t* q8_out;         // EMIT kernels: the result again as e4m3 [M][ld_q8] ...
  uint32_t* q8_scale;      // ... with its block scales, uint32 [ceil(columns / 128)][M]
  int ld_q8;
  int q8_fixed_col;        // EMIT kernels: result columns >= this are quantised with the FIXED multiplier below (clamped to +-448, scale byte
  float q8_fixed_mul;      //   127) instead of per-block scales — the v third of a q|k|v projection, whose P V contraction runs over keys
  float* rowstat_out;      // EMIT kernels (optional): fp32 [N / wave columns][M][2] partial (sum, sum of squares) of the result's rows
};

struct C3Params {
  const uint16_t* a;
  const uint16_t* w;
  const float* bias;
  const uint16_t* res;
  const float* rowvec;
  uint16_t* out;
  int N, C;                // output / input channels (C a multiple of 64)
  int B, H, W;             // map (input = output size)
  int ldw, ldo, ldr, ldrv;
  float alpha;
  int tiles_x, tiles_y, tiles_m, tiles_n, n_block, tiles;
  int chunks, splitk, ch_per;
  unsigned a_bytes, w_bytes;
  int G;
  int* counters;
  float* slabs;
  float* colstats;         // STATS kernels: fp32 [tiles_m * 2][N][2] per-(wave pixel block, channel) (sum, sum of squares)
  int geo, tw, th;         // host side: kernel instance (0: 16x8, 1: 8x8, 2: 16x8 upsampling, 3: wide.h 16x16 x 160 channels) and its pixel tile
  int bn, wgm;             // host side: output channels per tile, wave pixel blocks per tile (statistics slots)
  int dbg;                 // measurement builds only (UDT_DBG): bit 0 no weight DMA, 1 no patch DMA, 2 no MFMA, 3 no LDS fragment reads
};

}  // namespace lg
