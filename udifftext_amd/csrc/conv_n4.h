// conv_n4.h — 3x3 / stride 1 / pad 1 convolution with FOUR (padded) output channels and fp32 output (included by gemm.hip, round 6).
//
// The UNet's last layer (reference openaimodel.py:486-490 `out`: GroupNorm, SiLU, conv_nd(model_channels -> 4)) and the VAE decoder's
// conv_out (model.py:586-588, 128 -> 3) have 4 output channels: as an implicit GEMM on the gathered 256 x 64 tile their N is padded
// to 64 and every A fragment is gathered from global memory (47 us per UNet call at 64 x 64 x 320 for 0.38 GFLOP, 300 us on the VAE's
// 512 x 512 x 128 map), while the layer is one read of its input: 21 MB.  Here:
//   * a workgroup owns an 8 x 8 pixel tile of one sample: its 10 x 10 x C halo goes to LDS once (16-byte pieces, eight loads per
//     thread in flight; pixel pitch C + 8 elements: the b128 fragment reads of 16 neighbouring pixels fall on different banks), the
//     weights beside it as [4][9 C];
//   * wave q of the four takes channel quarter q for all 64 pixels: per tap and 16 channels two v_mfma_f32_32x32x16_bf16 (pixels
//     0-31 / 32-63 as the A rows, the 4 output channels as B columns 0-3, columns 4-31 zero: the matrix pipe is 8x over-provisioned
//     for this layer and still 6x faster than v_dot2_f32_bf16 on the vector ALU, which runs at quarter rate — the first form of this
//     kernel: 33 us);
//   * the four quarters of a pixel are added through LDS in a fixed order; + bias; one 16-byte fp32 store per pixel.
#pragma once

namespace cn4 {

struct Params {
  const uint16_t* a;       // bf16 NHWC [B][H][W][C]
  const uint16_t* w;       // bf16 [4][ldw], column k = tap * C + c (udifftext_amd.packing.pack_conv)
  const float* bias;       // [4] or nullptr
  float* out;              // fp32 [B * H * W][ldo]
  int B, H, W, C, ldw, ldo, tiles_x, tiles_y;
};

UDT_DEVINL int halo_bytes(int C) { return 100 * (C + 8) * 2; }

__global__ void __launch_bounds__(256) conv3x3_n4_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) char cn4_smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int pitch = p.C + 8;                                           // elements per halo pixel (16-byte multiple)
  uint16_t* halo = reinterpret_cast<uint16_t*>(cn4_smem);              // [100][pitch]
  uint16_t* wl = halo + 100 * pitch;                                   // [4][9 C]
  int tile = blockIdx.x;
  const int b = tile / (p.tiles_x * p.tiles_y);
  tile -= b * p.tiles_x * p.tiles_y;
  const int ty0 = (tile / p.tiles_x) * 8, tx0 = (tile - (tile / p.tiles_x) * p.tiles_x) * 8;
  const int c8 = p.C >> 3;
  // ---- halo and weights -> LDS (eight / six independent 16-byte loads per thread in flight: at C = 320 one workgroup fits a CU)
  for (int i0 = tid; i0 < 100 * c8; i0 += 256 * 8) {
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = i0 + 256 * j;
      const int px = i / c8, ch = i - px * c8;
      const int y = ty0 + px / 10 - 1, x = tx0 + px % 10 - 1;
      v[j] = u32x4{0u, 0u, 0u, 0u};
      if (i < 100 * c8 && y >= 0 && y < p.H && x >= 0 && x < p.W)
        v[j] = *reinterpret_cast<const u32x4*>(p.a + (((long long)b * p.H + y) * p.W + x) * p.C + ch * 8);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = i0 + 256 * j;
      if (i < 100 * c8) {
        const int px = i / c8, ch = i - px * c8;
        *reinterpret_cast<u32x4*>(halo + px * pitch + ch * 8) = v[j];
      }
    }
  }
  {
    const int nq = 9 * c8;                                             // 16-byte pieces of one output channel's row
    for (int i0 = tid; i0 < 4 * nq; i0 += 256 * 6) {
      u32x4 v[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int i = i0 + 256 * j, o = i / nq, qd = i - o * nq;
        v[j] = u32x4{0u, 0u, 0u, 0u};
        if (i < 4 * nq) v[j] = *reinterpret_cast<const u32x4*>(p.w + (long long)o * p.ldw + qd * 8);
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int i = i0 + 256 * j;
        if (i < 4 * nq) *reinterpret_cast<u32x4*>(wl + i * 8) = v[j];
      }
    }
  }
  __syncthreads();
  const int cq = __builtin_amdgcn_readfirstlane(tid >> 6);             // channel quarter of this wave
  const int cper = p.C >> 2;                                           // channels per quarter (a multiple of 16)
  f32x16 acc[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
  // A fragment of pixel block rb: lane (l31, hi) = pixel 32 rb + l31, channels c .. c + 7 with c = quarter base + 16 ks + 8 hi
  const int pa0 = ((l31 >> 3) * 10 + (l31 & 7)) * pitch, pa1 = pa0 + 40 * pitch;   // (block 1: four rows further down)
  const int cbase = cq * cper + hi * 8;
  const bf16x8_t zero = __builtin_bit_cast(bf16x8_t, u32x4{0u, 0u, 0u, 0u});
  for (int tap = 0; tap < 9; ++tap) {
    const int toff = ((tap / 3) * 10 + tap % 3) * pitch + cbase;
    const uint16_t* wrow = wl + (l31 & 3) * 9 * p.C + tap * p.C + cbase;
    for (int ks = 0; ks < cper; ks += 16) {
      const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(halo + pa0 + toff + ks);
      const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(halo + pa1 + toff + ks);
      bf16x8_t bw = *reinterpret_cast<const bf16x8_t*>(wrow + ks);
      if (l31 >= 4) bw = zero;                                        // B columns 4 .. 31: no such output channel
      // mfma32(A, B): acc[r] of lane (l31, hi) = sum_kk A[i = 8 (r >> 2) + 4 hi + (r & 3)][kk] B[j = l31][kk]  (common.h)
      acc[0] = mfma32(a0, bw, acc[0]);
      acc[1] = mfma32(a1, bw, acc[1]);
    }
  }
  __syncthreads();                                                     // the halo becomes the reduction scratch
  float* red = reinterpret_cast<float*>(cn4_smem);                     // [4 quarters][64 pixels][4]
  if (l31 < 4) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pixel = rb * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
        red[(cq * 64 + pixel) * 4 + l31] = acc[rb][r];
      }
  }
  __syncthreads();
  if (tid < 64) {
    const int y = ty0 + (tid >> 3), x = tx0 + (tid & 7);
    if (y < p.H && x < p.W) {
      f32x4 s = *reinterpret_cast<const f32x4*>(red + tid * 4);
#pragma unroll
      for (int q = 1; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(red + (q * 64 + tid) * 4);
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
      }
      if (p.bias) { s[0] += p.bias[0]; s[1] += p.bias[1]; s[2] += p.bias[2]; s[3] += p.bias[3]; }
      *reinterpret_cast<f32x4*>(p.out + (((long long)b * p.H + y) * p.W + x) * p.ldo) = s;
    }
  }
}

}  // namespace cn4
