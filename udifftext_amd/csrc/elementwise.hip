// elementwise.hip — sampler-step math and layout changes at the NCHW fp32 plugin boundary (HBM-bound).
//
//   udt_unet_input / udt_cfg_euler_step : guiders.py:25-40, denoiser.py:22-28, denoiser_scaling.py:16-22,
//                                         sampling_utils.py:8-9,39-40, sampling.py:85-86,348-351
//   udt_posterior_sample                : distributions.py:24-41 (+ LatentEncoder scale, encoders/modules.py:1011-1014)
//   udt_embed_tokens                    : encoders/modules.py:1069-1085,1160-1163
//   udt_timestep_embedding              : diffusionmodules/util.py:206-230
//   udt_mask_downsample                 : encoders/modules.py:843-857 (bilinear x0.125, align_corners False)
//   udt_local_loss                      : diffusionmodules/loss.py:192-235
#include "common.h"

namespace {

__global__ void unet_input_kernel(const float* __restrict__ x, uint16_t* __restrict__ xin, int B, int hw, int cpad,
                                  float c_in) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // (b, pixel)
  if (i >= B * hw) return;
  const int b = i / hw;
  const int pix = i - b * hw;
  const float* xb = x + (long long)b * 4 * hw + pix;
  const float v0 = xb[0] * c_in, v1 = xb[hw] * c_in, v2 = xb[2 * hw] * c_in, v3 = xb[3 * hw] * c_in;
  const u32x2 pk = {pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
  *reinterpret_cast<u32x2*>(xin + ((long long)b * hw + pix) * cpad) = pk;
  *reinterpret_cast<u32x2*>(xin + ((long long)(b + B) * hw + pix) * cpad) = pk;
}

__global__ void cfg_euler_kernel(float* __restrict__ x, const float* __restrict__ eps, float* __restrict__ den_out,
                                 int B, int hw, int ld, float c_out, float sigma, float sigma_next, float scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * hw) return;
  const int b = i / hw;
  const int pix = i - b * hw;
  const f32x4 eu = *reinterpret_cast<const f32x4*>(eps + ((long long)b * hw + pix) * ld);
  const f32x4 ec = *reinterpret_cast<const f32x4*>(eps + ((long long)(b + B) * hw + pix) * ld);
  float* xb = x + (long long)b * 4 * hw + pix;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float xv = xb[c * hw];
    const float du = eu[c] * c_out + xv;           // network(...)*c_out + input*c_skip  (denoiser.py:28)
    const float dc = ec[c] * c_out + xv;
    const float den = du + scale * (dc - du);         // sampling_utils.py:8-9
    const float d = (xv - den) / sigma;               // to_d, sampling_utils.py:39-40
    xb[c * hw] = xv + d * (sigma_next - sigma);       // euler_step, sampling.py:85-86
    if (den_out) den_out[(long long)b * 4 * hw + c * hw + pix] = den;
  }
}

__global__ void posterior_sample_kernel(const float* __restrict__ mom, const float* __restrict__ noise,
                                        float* __restrict__ z, int B, int hw, int ldm, float scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * hw) return;
  const int b = i / hw;
  const int pix = i - b * hw;
  const float* m = mom + ((long long)b * hw + pix) * ldm;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float mean = m[c];
    float logvar = m[4 + c];
    logvar = fminf(fmaxf(logvar, -30.0f), 20.0f);
    const float stdv = expf(0.5f * logvar);
    const long long o = (long long)b * 4 * hw + c * hw + pix;
    z[o] = scale * (mean + stdv * noise[o]);
  }
}

// one thread per (b, pixel, 8-channel chunk)
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int B, int C,
                                    long long HW, int cpad, float scale) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c8 = cpad >> 3;
  const long long total = (long long)B * HW * c8;
  if (i >= total) return;
  const int cc = (int)(i % c8);
  const long long bp = i / c8;
  const long long b = bp / HW;
  const long long pix = bp - b * HW;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cc * 8 + j;
    v[j] = (c < C) ? x[(b * C + c) * HW + pix] * scale : 0.f;
  }
  u32x4 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
  *reinterpret_cast<u32x4*>(y + bp * cpad + cc * 8) = pk;
}

__global__ void nhwc_to_nchw_kernel(const void* __restrict__ x, float* __restrict__ y, int B, int C, long long HW,
                                    int ld, int src_f32) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (b, c, pix), pix fastest
  const long long total = (long long)B * C * HW;
  if (i >= total) return;
  const long long pix = i % HW;
  const long long bc = i / HW;
  const int c = (int)(bc % C);
  const long long b = bc / C;
  const long long src = (b * HW + pix) * ld + c;
  y[i] = src_f32 ? reinterpret_cast<const float*>(x)[src]
                 : bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(x)[src]);
}

__global__ void nhwc_set_channels_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int B, int C,
                                         long long HW, int cpad, int c0) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (b, pix)
  if (i >= (long long)B * HW) return;
  const long long b = i / HW;
  const long long pix = i - b * HW;
  for (int c = 0; c < C; ++c) {
    const float v = src[(b * C + c) * HW + pix];
    dst[i * cpad + c0 + c] = (uint16_t)(pack_bf16x2(v, 0.f) & 0xffffu);
  }
}

__global__ void embed_tokens_kernel(const int32_t* __restrict__ idx, const float* __restrict__ table,
                                    const float* __restrict__ pe, uint16_t* __restrict__ out, int n_tok, int L,
                                    int D) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (token, d/2)
  const int d2 = D >> 1;
  if (i >= (long long)n_tok * d2) return;
  const int tok = (int)(i / d2);
  const int d = (int)(i - (long long)tok * d2) * 2;
  const int id = idx[tok];
  const int pos = tok % L;
  const float a = table[(long long)id * D + d] + pe[(long long)pos * D + d];
  const float b = table[(long long)id * D + d + 1] + pe[(long long)pos * D + d + 1];
  *reinterpret_cast<uint32_t*>(out + (long long)tok * D + d) = pack_bf16x2(a, b);
}

__global__ void timestep_embedding_kernel(const float* __restrict__ t, uint16_t* __restrict__ out, int n, int dim) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // (sample, k) with k < dim/2
  const int half = dim >> 1;
  if (i >= n * half) return;
  const int s = i / half;
  const int k = i - s * half;
  const float freq = expf(-9.210340371976184f * (float)k / (float)half);   // ln(10000)
  const float arg = t[s] * freq;
  out[(long long)s * dim + k] = (uint16_t)(pack_bf16x2(cosf(arg), 0.f) & 0xffffu);
  out[(long long)s * dim + half + k] = (uint16_t)(pack_bf16x2(sinf(arg), 0.f) & 0xffffu);
}

__global__ void mask_downsample_kernel(const float* __restrict__ m, float* __restrict__ out, int B, int H, int W) {
  const int h = H >> 3, w = W >> 3;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * h * w) return;
  const int b = i / (h * w);
  const int r = i - b * h * w;
  const int oy = r / w, ox = r - (r / w) * w;
  const float* src = m + (long long)b * H * W;
  const int y0 = oy * 8 + 3, x0 = ox * 8 + 3;
  // bilinear, scale 1/8, align_corners False: source coordinate 8*o + 3.5 -> equal weights on the centre 2x2
  out[i] = 0.25f * (src[y0 * W + x0] + src[y0 * W + x0 + 1] + src[(y0 + 1) * W + x0] + src[(y0 + 1) * W + x0 + 1]);
}

// one workgroup per sample; the head-averaged map of one token (size x size floats, size <= 128) lives in LDS
__global__ void __launch_bounds__(256) local_loss_kernel(const float* __restrict__ probs, const float* __restrict__ mask,
                                                         const float* __restrict__ seg, const float* __restrict__ gk,
                                                         float* __restrict__ loss, int heads, int size, int L,
                                                         int seg_l, int Hm, int Wm, int mask_batch) {
  extern __shared__ __attribute__((aligned(16))) float llsm[];   // [size*size] map, [4] wave maxima, [1] best
  float* amap = llsm;
  float* red = llsm + size * size;
  float& best = red[4];
  const int b = blockIdx.x;
  const int bm = b % mask_batch;                   // sample b is scored against mask / seg row b % mask_batch (tiled candidates)
  const int t = threadIdx.x;
  const int n = size * size;
  if (t == 0) best = INFINITY;
  __syncthreads();
  for (int l = 0; l < seg_l; ++l) {
    for (int i = t; i < n; i += 256) {
      float a = 0.f;
      for (int h = 0; h < heads; ++h) a += probs[(((long long)b * heads + h) * n + i) * L + l];
      amap[i] = a / (float)heads;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int i = t; i < n; i += 256) {
      const int y = i / size, x = i - (i / size) * size;
      float acc = 0.f;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const int yy = y + dy, xx = x + dx;
          if (yy >= 0 && yy < size && xx >= 0 && xx < size) acc += gk[(dy + 1) * 3 + dx + 1] * amap[yy * size + xx];
        }
      // F.interpolate(mask, (size,size)) nearest: src = floor(dst * in / out)
      const int my = (int)(((long long)y * Hm) / size), mxx = (int)(((long long)x * Wm) / size);
      const float mv = mask[((long long)bm * Hm + my) * Wm + mxx];
      mx = fmaxf(mx, mv * acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    if (t == 0) {
      const float m4 = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
      const float pl = m4 + (1.0f - seg[(long long)bm * seg_l + l]);
      best = fminf(best, pl);
    }
    __syncthreads();
  }
  if (t == 0) loss[b] += -best;
}

__global__ void add_bf16_kernel(uint16_t* __restrict__ x, const uint16_t* __restrict__ y, long long n8) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  u32x4 a = *reinterpret_cast<const u32x4*>(x + i * 8);
  const u32x4 b = *reinterpret_cast<const u32x4*>(y + i * 8);
#pragma unroll
  for (int j = 0; j < 4; ++j) a[j] = pack_bf16x2(bf16_lo(a[j]) + bf16_lo(b[j]), bf16_hi(a[j]) + bf16_hi(b[j]));
  *reinterpret_cast<u32x4*>(x + i * 8) = a;
}

// out[r][c] = x[r][c] + bias[c]  (bf16 rows, fp32 bias, 8 channels per thread)
__global__ void bias_add_kernel(const uint16_t* __restrict__ x, const float* __restrict__ bias, uint16_t* __restrict__ out,
                                long long n8, int c8) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const int c = (int)(i % c8) * 8;
  u32x4 a = *reinterpret_cast<const u32x4*>(x + i * 8);
  const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + c);
  const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + c + 4);
  a[0] = pack_bf16x2(bf16_lo(a[0]) + b0[0], bf16_hi(a[0]) + b0[1]);
  a[1] = pack_bf16x2(bf16_lo(a[1]) + b0[2], bf16_hi(a[1]) + b0[3]);
  a[2] = pack_bf16x2(bf16_lo(a[2]) + b1[0], bf16_hi(a[2]) + b1[1]);
  a[3] = pack_bf16x2(bf16_lo(a[3]) + b1[2], bf16_hi(a[3]) + b1[3]);
  *reinterpret_cast<u32x4*>(out + i * 8) = a;
}

inline unsigned nblk(long long n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

}  // namespace

#define UDT_STREAM hipStream_t s = reinterpret_cast<hipStream_t>(stream); UdtProfScope prof(5, s)

extern "C" int udt_unet_input(const float* x, void* xin, int32_t B, int32_t hw, int32_t cpad, float c_in,
                              void* stream) {
  if (!x || !xin) return UDT_ERR_BAD_ARG;
  if (B <= 0 || hw <= 0 || cpad < 8 || cpad % 8 != 0) return UDT_ERR_BAD_SHAPE;
  UDT_STREAM;
  hipLaunchKernelGGL(unet_input_kernel, dim3(nblk((long long)B * hw)), dim3(256), 0, s, x,
                     reinterpret_cast<uint16_t*>(xin), B, hw, cpad, c_in);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_cfg_euler_step(float* x, const float* eps, float* denoised_out, int32_t B, int32_t hw,
                                  int32_t ld_eps, float c_out, float sigma, float sigma_next, float cfg_scale, void* stream) {
  if (!x || !eps) return UDT_ERR_BAD_ARG;
  if (B <= 0 || hw <= 0 || ld_eps < 4 || ld_eps % 4 != 0 || sigma == 0.f) return UDT_ERR_BAD_SHAPE;
  UDT_STREAM;
  hipLaunchKernelGGL(cfg_euler_kernel, dim3(nblk((long long)B * hw)), dim3(256), 0, s, x, eps, denoised_out, B, hw,
                     ld_eps, c_out, sigma, sigma_next, cfg_scale);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_posterior_sample(const float* moments, const float* noise, float* z, int32_t B, int32_t hw,
                                    int32_t ldm, float scale, void* stream) {
  if (!moments || !noise || !z) return UDT_ERR_BAD_ARG;
  if (B <= 0 || hw <= 0 || ldm < 8) return UDT_ERR_BAD_SHAPE;
  UDT_STREAM;
  hipLaunchKernelGGL(posterior_sample_kernel, dim3(nblk((long long)B * hw)), dim3(256), 0, s, moments, noise, z, B,
                     hw, ldm, scale);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_nchw_to_nhwc(const float* x, void* y, int32_t B, int32_t C, int64_t HW, int32_t cpad, float scale,
                                void* stream) {
  if (!x || !y) return UDT_ERR_BAD_ARG;
  if (B <= 0 || C <= 0 || HW <= 0 || cpad < C || cpad % 8 != 0) return UDT_ERR_BAD_SHAPE;
  UDT_STREAM;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(nblk((long long)B * HW * (cpad / 8))), dim3(256), 0, s, x,
                     reinterpret_cast<uint16_t*>(y), B, C, (long long)HW, cpad, scale);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_nhwc_to_nchw(const void* x, float* y, int32_t B, int32_t C, int64_t HW, int32_t ld,
                                int32_t src_is_f32, void* stream) {
  if (!x || !y) return UDT_ERR_BAD_ARG;
  if (B <= 0 || C <= 0 || HW <= 0 || ld < C) return UDT_ERR_BAD_SHAPE;
  UDT_STREAM;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(nblk((long long)B * C * HW)), dim3(256), 0, s, x, y, B, C,
                     (long long)HW, ld, src_is_f32);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_nhwc_set_channels(const float* src, void* dst, int32_t B, int32_t C, int64_t HW, int32_t cpad,
                                     int32_t c0, void* stream) {
  if (!src || !dst) return UDT_ERR_BAD_ARG;
  if (B <= 0 || C <= 0 || HW <= 0 || c0 < 0 || c0 + C > cpad) return UDT_ERR_BAD_SHAPE;
  UDT_STREAM;
  hipLaunchKernelGGL(nhwc_set_channels_kernel, dim3(nblk((long long)B * HW)), dim3(256), 0, s, src,
                     reinterpret_cast<uint16_t*>(dst), B, C, (long long)HW, cpad, c0);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_embed_tokens(const int32_t* idx, const float* table, const float* pe, void* out, int32_t n_tok,
                                int32_t L, int32_t D, void* stream) {
  if (!idx || !table || !pe || !out) return UDT_ERR_BAD_ARG;
  if (n_tok <= 0 || L <= 0 || D <= 0 || D % 2 != 0) return UDT_ERR_BAD_SHAPE;
  UDT_STREAM;
  hipLaunchKernelGGL(embed_tokens_kernel, dim3(nblk((long long)n_tok * (D / 2))), dim3(256), 0, s, idx, table, pe,
                     reinterpret_cast<uint16_t*>(out), n_tok, L, D);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_timestep_embedding(const float* t, void* out, int32_t n, int32_t dim, void* stream) {
  if (!t || !out) return UDT_ERR_BAD_ARG;
  if (n <= 0 || dim <= 0 || dim % 2 != 0) return UDT_ERR_BAD_SHAPE;
  UDT_STREAM;
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(nblk((long long)n * (dim / 2))), dim3(256), 0, s, t,
                     reinterpret_cast<uint16_t*>(out), n, dim);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_mask_downsample(const float* mask, float* out, int32_t B, int32_t H, int32_t W, void* stream) {
  if (!mask || !out) return UDT_ERR_BAD_ARG;
  if (B <= 0 || H <= 0 || W <= 0 || H % 8 != 0 || W % 8 != 0) return UDT_ERR_BAD_SHAPE;
  UDT_STREAM;
  hipLaunchKernelGGL(mask_downsample_kernel, dim3(nblk((long long)B * (H / 8) * (W / 8))), dim3(256), 0, s, mask,
                     out, B, H, W);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

static int local_loss_impl(const float* probs, const float* mask, const float* seg_mask, const float* gkernel9, float* loss_accum,
                           int32_t n_samples, int32_t mask_batch, int32_t heads, int32_t size, int32_t L, int32_t seg_l, int32_t Hm,
                           int32_t Wm, void* stream) {
  if (!probs || !mask || !seg_mask || !gkernel9 || !loss_accum) return UDT_ERR_BAD_ARG;
  if (n_samples <= 0 || mask_batch <= 0 || n_samples % mask_batch != 0 || heads <= 0 || size <= 0 || size > 120 || L <= 0 ||
      seg_l <= 0 || seg_l > L) return UDT_ERR_BAD_SHAPE;
  UDT_STREAM;
  // 96x96 maps (768x768 inputs) need 36 KiB; the default dynamic-LDS limit is 64 KiB (120x120 floats + 32 B fit)
  const size_t smem = ((size_t)size * size + 8) * sizeof(float);
  hipLaunchKernelGGL(local_loss_kernel, dim3(n_samples), dim3(256), smem, s, probs, mask, seg_mask, gkernel9, loss_accum, heads,
                     size, L, seg_l, Hm, Wm, mask_batch);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_local_loss(const float* probs, const float* mask, const float* seg_mask, const float* gkernel9,
                              float* loss_accum, int32_t B, int32_t heads, int32_t size, int32_t L, int32_t seg_l,
                              int32_t Hm, int32_t Wm, void* stream) {
  return local_loss_impl(probs, mask, seg_mask, gkernel9, loss_accum, B, B, heads, size, L, seg_l, Hm, Wm, stream);
}

extern "C" int udt_local_loss_tiled(const float* probs, const float* mask, const float* seg_mask, const float* gkernel9,
                                    float* loss_accum, int32_t n_samples, int32_t mask_batch, int32_t heads, int32_t size, int32_t L,
                                    int32_t seg_l, int32_t Hm, int32_t Wm, void* stream) {
  return local_loss_impl(probs, mask, seg_mask, gkernel9, loss_accum, n_samples, mask_batch, heads, size, L, seg_l, Hm, Wm, stream);
}

extern "C" int udt_bias_add_bf16(const void* x, const float* bias, void* out, int64_t rows, int32_t C, void* stream) {
  if (!x || !bias || !out) return UDT_ERR_BAD_ARG;
  if (rows <= 0 || C <= 0 || C % 8 != 0) return UDT_ERR_BAD_SHAPE;
  UDT_STREAM;
  const long long n8 = rows * (C / 8);
  hipLaunchKernelGGL(bias_add_kernel, dim3(nblk(n8)), dim3(256), 0, s, reinterpret_cast<const uint16_t*>(x), bias,
                     reinterpret_cast<uint16_t*>(out), n8, C / 8);
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}

extern "C" int udt_add_bf16(void* x, const void* y, int64_t n, void* stream) {
  if (!x || !y) return UDT_ERR_BAD_ARG;
  if (n <= 0 || n % 8 != 0) return UDT_ERR_BAD_SHAPE;
  UDT_STREAM;
  hipLaunchKernelGGL(add_bf16_kernel, dim3(nblk(n / 8)), dim3(256), 0, s, reinterpret_cast<uint16_t*>(x),
                     reinterpret_cast<const uint16_t*>(y), (long long)(n / 8));
  UDT_CHECK_LAUNCH();
  return UDT_OK;
}
