// pack.hip — immutable packed-weight handles (the only allocations the library makes on a caller's behalf: an explicit
// udt_pack_* / udt_free_packed pair) and the SURVEY §8b names of the generic entry points.
//
// The checkpoint layouts are the reference's: nn.Linear weight [out, in] fp32, nn.Conv2d weight [out, in, kh, kw] fp32
// (sgm/modules/attention.py:47-99, sgm/modules/diffusionmodules/openaimodel.py:96-133,183-231).  The device layouts are the ones
// udt_gemm consumes: bf16 [Npad][Kpad] K-contiguous rows (K padded to 64 with zeros, N to a multiple of 4); convolutions
// tap-major k = (ky * kw + kx) * Cpad + c with every concatenated source padded to 64 channels separately; GEGLU projections
// row-permuted into [32 value | 32 gate] blocks; fp8 = OCP e4m3 bytes [Npad][Kpad128] + one fp32 scale per output channel
// (max_k |w| / 448).  udifftext_amd/packing.py is the same packing in torch (the Python front end uses that one); the GPU
// tests compare the two bit for bit.
#include "common.h"
#include <stdlib.h>

struct udt_packed {
  void* weight;
  float* bias;
  float* colscale;
  int32_t N, Npad, K, Kpad, dtype;
};

namespace {

constexpr float FP8_MAX = 448.0f;

// source row of packed row n: identity, or the GEGLU block permutation (packing.geglu_permutation)
UDT_DEVINL int src_row(int n, int N, int geglu) {
  if (!geglu) return n;
  const int inner = N >> 1, blk = n >> 6, r = n & 63;
  return r < 32 ? blk * 32 + r : inner + blk * 32 + (r - 32);
}

__global__ void __launch_bounds__(256) pack_linear_bf16_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int N, int K,
                                                               int Npad, int Kpad, int geglu) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // one packed PAIR of elements
  const long long pairs = (long long)Npad * (Kpad >> 1);
  if (i >= pairs) return;
  const int n = (int)(i / (Kpad >> 1)), k = (int)(i - (long long)n * (Kpad >> 1)) * 2;
  float a = 0.f, b = 0.f;
  if (n < N) {
    const float* row = w + (long long)src_row(n, N, geglu) * K;
    if (k < K) a = row[k];
    if (k + 1 < K) b = row[k + 1];
  }
  reinterpret_cast<uint32_t*>(out)[i] = pack_bf16x2(a, b);
}

// one workgroup per packed row: amax -> scale = amax / 448 -> e4m3 bytes
__global__ void __launch_bounds__(256) pack_linear_fp8_kernel(const float* __restrict__ w, uint8_t* __restrict__ out,
                                                              float* __restrict__ colscale, int N, int K, int Kpad, int geglu) {
  __shared__ float red[256];
  const int n = blockIdx.x, t = threadIdx.x;
  uint8_t* orow = out + (long long)n * Kpad;
  if (n >= N) {
    for (int k = t; k < Kpad; k += 256) orow[k] = 0;
    if (t == 0) colscale[n] = 1.0f;
    return;
  }
  const float* row = w + (long long)src_row(n, N, geglu) * K;
  float m = 0.f;
  for (int k = t; k < K; k += 256) m = fmaxf(m, fabsf(row[k]));
  red[t] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) red[t] = fmaxf(red[t], red[t + s]);
    __syncthreads();
  }
  // (amax * (1 / 448): what torch computes for a tensor divided by a scalar, so that the scales agree bit for bit with packing.py)
  const float scale = fmaxf(red[0], 1e-12f) * (1.0f / FP8_MAX);
  if (t == 0) colscale[n] = scale;
  for (int k = t * 2; k < Kpad; k += 512) {
    float a = 0.f, b = 0.f;
    if (k < K) a = fminf(fmaxf(row[k] / scale, -FP8_MAX), FP8_MAX);
    if (k + 1 < K) b = fminf(fmaxf(row[k + 1] / scale, -FP8_MAX), FP8_MAX);
    const int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    *reinterpret_cast<uint16_t*>(orow + k) = (uint16_t)(v & 0xffff);
  }
}

__global__ void __launch_bounds__(256) pack_bias_kernel(const float* __restrict__ b, float* __restrict__ out, int N, int Npad, int geglu) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= Npad) return;
  out[n] = (b && n < N) ? b[src_row(n, N, geglu)] : 0.f;
}

// [N, Cin, kh, kw] fp32 -> bf16 [Npad][kh*kw][sum of padded segments]; seg_src[c'] = source channel of packed channel c' or -1
__global__ void __launch_bounds__(256) pack_conv_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, const int* __restrict__ seg_src,
                                                        int N, int Cin, int taps, int Cpad, int Npad) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)Npad * taps * Cpad;
  if (i >= total) return;
  const int c = (int)(i % Cpad);
  const int tap = (int)((i / Cpad) % taps);
  const int n = (int)(i / ((long long)Cpad * taps));
  float v = 0.f;
  const int sc = seg_src[c];
  if (n < N && sc >= 0) v = w[((long long)n * Cin + sc) * taps + tap];
  out[i] = (uint16_t)(pack_bf16x2(v, 0.f) & 0xffff);
}

int alloc(void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes ? bytes : 16);
  return e == hipSuccess ? UDT_OK : udt_set_hip_error(e);
}

}  // namespace

extern "C" int udt_free_packed(udt_packed* h) {
  if (!h) return UDT_OK;
  hipError_t e = hipSuccess, e2;
  if (h->weight && (e2 = hipFree(h->weight)) != hipSuccess) e = e2;
  if (h->bias && (e2 = hipFree(h->bias)) != hipSuccess) e = e2;
  if (h->colscale && (e2 = hipFree(h->colscale)) != hipSuccess) e = e2;
  free(h);
  return e == hipSuccess ? UDT_OK : udt_set_hip_error(e);
}

extern "C" int udt_pack_linear(const float* w, const float* bias, int32_t N, int32_t K, int32_t dtype, int32_t geglu,
                               udt_packed** out, void* stream) {
  if (!w || !out) return UDT_ERR_BAD_ARG;
  if (N <= 0 || K <= 0 || (dtype != UDT_DTYPE_BF16 && dtype != UDT_DTYPE_FP8_E4M3)) return UDT_ERR_BAD_SHAPE;
  if (geglu && (N % 64 != 0)) return UDT_ERR_BAD_SHAPE;              // [32 value | 32 gate] row blocks
  udt_packed* h = static_cast<udt_packed*>(calloc(1, sizeof(udt_packed)));
  if (!h) return UDT_ERR_BAD_ARG;
  const bool f8 = dtype == UDT_DTYPE_FP8_E4M3;
  h->N = N; h->K = K; h->dtype = dtype;
  h->Npad = (N + 3) / 4 * 4;
  h->Kpad = f8 ? (K + 127) / 128 * 128 : (K + 63) / 64 * 64;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rc = alloc(&h->weight, (size_t)h->Npad * h->Kpad * (f8 ? 1 : 2));
  if (rc == UDT_OK) rc = alloc(reinterpret_cast<void**>(&h->bias), (size_t)h->Npad * sizeof(float));
  if (rc == UDT_OK && f8) rc = alloc(reinterpret_cast<void**>(&h->colscale), (size_t)h->Npad * sizeof(float));
  if (rc != UDT_OK) { udt_free_packed(h); return rc; }
  if (f8) {
    hipLaunchKernelGGL(pack_linear_fp8_kernel, dim3(h->Npad), dim3(256), 0, s, w, static_cast<uint8_t*>(h->weight), h->colscale, N, K,
                       h->Kpad, geglu ? 1 : 0);
  } else {
    const long long pairs = (long long)h->Npad * (h->Kpad / 2);
    hipLaunchKernelGGL(pack_linear_bf16_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s, w,
                       static_cast<uint16_t*>(h->weight), N, K, h->Npad, h->Kpad, geglu ? 1 : 0);
  }
  hipLaunchKernelGGL(pack_bias_kernel, dim3((h->Npad + 255) / 256), dim3(256), 0, s, bias, h->bias, N, h->Npad, geglu ? 1 : 0);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { udt_free_packed(h); return udt_set_hip_error(e); }
  *out = h;
  return UDT_OK;
}

extern "C" int udt_pack_conv(const float* w, const float* bias, int32_t N, int32_t Cin, int32_t kh, int32_t kw, const int32_t* segments,
                             int32_t n_segments, int32_t n_pad_to, udt_packed** out, void* stream) {
  if (!w || !out) return UDT_ERR_BAD_ARG;
  if (N <= 0 || Cin <= 0 || kh <= 0 || kw <= 0 || n_pad_to <= 0 || n_segments < 0 || (n_segments > 0 && !segments)) return UDT_ERR_BAD_SHAPE;
  // packed channel -> source channel (every concatenated source is padded to a multiple of 64 channels separately)
  int Cpad = 0, sum = 0;
  const int one = Cin;
  const int32_t* segs = n_segments > 0 ? segments : &one;
  const int ns = n_segments > 0 ? n_segments : 1;
  for (int i = 0; i < ns; ++i) {
    if (segs[i] <= 0) return UDT_ERR_BAD_SHAPE;
    Cpad += (segs[i] + 63) / 64 * 64;
    sum += segs[i];
  }
  if (sum != Cin) return UDT_ERR_BAD_SHAPE;
  int* map = static_cast<int*>(malloc((size_t)Cpad * sizeof(int)));
  udt_packed* h = static_cast<udt_packed*>(calloc(1, sizeof(udt_packed)));
  if (!map || !h) { free(map); free(h); return UDT_ERR_BAD_ARG; }
  for (int i = 0, c = 0, src = 0; i < ns; ++i) {
    const int sp = (segs[i] + 63) / 64 * 64;
    for (int j = 0; j < sp; ++j) map[c++] = j < segs[i] ? src + j : -1;
    src += segs[i];
  }
  const int taps = kh * kw;
  h->N = N; h->K = taps * Cpad; h->Kpad = h->K; h->dtype = UDT_DTYPE_BF16;
  h->Npad = (N + n_pad_to - 1) / n_pad_to * n_pad_to;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int* dmap = nullptr;
  int rc = alloc(&h->weight, (size_t)h->Npad * h->Kpad * 2);
  if (rc == UDT_OK) rc = alloc(reinterpret_cast<void**>(&h->bias), (size_t)h->Npad * sizeof(float));
  if (rc == UDT_OK) rc = alloc(reinterpret_cast<void**>(&dmap), (size_t)Cpad * sizeof(int));
  if (rc == UDT_OK) {
    hipError_t e = hipMemcpyAsync(dmap, map, (size_t)Cpad * sizeof(int), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) rc = udt_set_hip_error(e);
  }
  if (rc == UDT_OK) {
    const long long total = (long long)h->Npad * h->Kpad;
    hipLaunchKernelGGL(pack_conv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, static_cast<uint16_t*>(h->weight), dmap, N,
                       Cin, taps, Cpad, h->Npad);
    hipLaunchKernelGGL(pack_bias_kernel, dim3((h->Npad + 255) / 256), dim3(256), 0, s, bias, h->bias, N, h->Npad, 0);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);              // the channel map is freed below (load-time call, not on the hot path)
    if (e != hipSuccess) rc = udt_set_hip_error(e);
  }
  if (dmap) (void)hipFree(dmap);
  free(map);
  if (rc != UDT_OK) { udt_free_packed(h); return rc; }
  *out = h;
  return UDT_OK;
}

extern "C" const void* udt_packed_weight(const udt_packed* h) { return h ? h->weight : nullptr; }
extern "C" const float* udt_packed_bias(const udt_packed* h) { return h ? h->bias : nullptr; }
extern "C" const float* udt_packed_colscale(const udt_packed* h) { return h ? h->colscale : nullptr; }
extern "C" int32_t udt_packed_dim(const udt_packed* h, int32_t which) {
  if (!h) return 0;
  switch (which) {
    case 0: return h->N;
    case 1: return h->Npad;
    case 2: return h->K;
    case 3: return h->Kpad;
    case 4: return h->dtype;
    default: return 0;
  }
}

// ---- SURVEY §8b names of the generic entry points ------------------------------------------------------------------------
extern "C" int udt_gemm_fwd(const udt_gemm_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
  return udt_gemm(d, workspace, workspace_bytes, stream);
}
extern "C" int udt_conv1x1_fwd(const udt_gemm_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
  if (!d || !(d->flags & UDT_GEMM_CONV) || d->ksize != 1) return UDT_ERR_BAD_ARG;
  return udt_gemm(d, workspace, workspace_bytes, stream);
}
extern "C" size_t udt_workspace_bytes(const udt_gemm_desc* d) { return udt_gemm_workspace_bytes(d); }
extern "C" int udt_sampler_step(float* x, const float* eps, float* denoised_out, int32_t B, int32_t hw, int32_t ld_eps, float c_out,
                                float sigma, float sigma_next, float cfg_scale, void* stream) {
  return udt_cfg_euler_step(x, eps, denoised_out, B, hw, ld_eps, c_out, sigma, sigma_next, cfg_scale, stream);
}
