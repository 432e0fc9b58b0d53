/*
 * udt_kernels.h — C ABI of libudt_kernels.so: the MI355X (gfx950 / CDNA4) kernels behind the
 * UDiffText denoising hot path.
 *
 * The reference (ZYM-PKU/UDiffText) is pure Python and owns no native boundary; its hot path reaches
 * native arithmetic only through third-party wheels.  Each entry point below therefore replaces the
 * third-party call the reference makes at the cited file:line (paths relative to the reference root):
 *
 *   udt_gemm            nn.Linear / nn.Conv2d(1x1) / nn.Conv2d(3x3) call sites:
 *                         sgm/modules/attention.py:47-51,66-70 (GEGLU / FeedForward),
 *                         sgm/modules/attention.py:193-199,215-218 (to_q/to_k/to_v/to_out),
 *                         sgm/modules/attention.py:375,395,407,411 (proj_in / proj_out),
 *                         sgm/modules/diffusionmodules/openaimodel.py:186,223-230,240 (ResBlock convs + skip),
 *                         sgm/modules/diffusionmodules/openaimodel.py:85-101,132-146 (Up/Downsample),
 *                         sgm/modules/diffusionmodules/openaimodel.py:210-216,340-344 (emb_layers, time_embed),
 *                         sgm/modules/diffusionmodules/model.py:59-68,74-85,108-124,128-148 (VAE convs),
 *                         sgm/modules/diffusionmodules/model.py:214-226,232-256 (VAE attention 1x1 convs),
 *                         sgm/models/autoencoder.py:297-298 (quant_conv / post_quant_conv),
 *                         sgm/modules/encoders/modules.py:1103-1104 (nn.TransformerEncoderLayer linears)
 *   udt_attn_fwd        xformers.ops.memory_efficient_attention  sgm/modules/attention.py:246
 *   udt_xattn_fwd       einsum / softmax / einsum               sgm/modules/attention.py:152-172
 *   udt_mattn_fwd       nn.MultiheadAttention (OCR scorer decoder)  src/parseq/strhub/models/parseq/modules.py:57-70
 *                       (also nn.MultiheadAttention inside nn.TransformerEncoderLayer,
 *                        sgm/modules/encoders/modules.py:1103-1104,1164)
 *   udt_softmax_rows    softmax of the VAE single-head attention  sgm/modules/diffusionmodules/model.py:246
 *   udt_gn_stats / udt_gn_apply
 *                       GroupNorm32 + SiLU      sgm/modules/diffusionmodules/util.py:258-275,
 *                                               sgm/modules/diffusionmodules/openaimodel.py:183-187,218-221,536-538
 *                       GroupNorm(eps 1e-6)     sgm/modules/attention.py:82-85,402 ; model.py:48-52,128-143
 *   udt_layernorm       nn.LayerNorm            sgm/modules/attention.py:297,310-311,315-339
 *   udt_unet_input / udt_cfg_euler_step
 *                       guider.prepare_inputs + denoiser scaling + CFG + Euler update
 *                       sgm/modules/diffusionmodules/guiders.py:25-40, denoiser.py:22-28,
 *                       denoiser_scaling.py:16-22, sampling_utils.py:8-9,39-40, sampling.py:85-86,348-351
 *   udt_posterior_sample  DiagonalGaussianDistribution.sample  sgm/modules/distributions/distributions.py:24-41
 *   udt_nchw_to_nhwc / udt_nhwc_to_nchw   layout change at the NCHW fp32 plugin boundary
 *                       (rearrange "b c h w -> b (h w) c", sgm/modules/attention.py:405,412)
 *   udt_embed_tokens    nn.Embedding + PositionalEncoding  sgm/modules/encoders/modules.py:1069-1085,1160-1163
 *   udt_timestep_embedding  timestep_embedding  sgm/modules/diffusionmodules/util.py:206-230
 *   udt_mask_downsample SpatialRescaler (bilinear x0.125)  sgm/modules/encoders/modules.py:843-857
 *   udt_local_loss_maps head-mean + 3x3 blur + masked max of t_attn maps  sgm/modules/diffusionmodules/loss.py:192-235
 *   the reverse pass (round 6; torch.autograd under attend_and_excite, sgm/modules/diffusionmodules/sampling.py:233-252, and under
 *   DiffusionEngine.training_step, sgm/models/diffusion.py:138-172 with loss.py:131-176,237-286):
 *     udt_attn_bwd, udt_xattn_bwd, udt_xattn_bwd_kv       autograd of the two attention call sites above
 *     udt_gn_bwd, udt_layernorm_bwd, udt_ln_param_grad, udt_geglu_fwd / _bwd   autograd of the norms / GEGLU
 *     udt_local_loss_bwd, udt_local_loss_seg_bwd, udt_diff_loss_grad          FullLoss.get_min_local_loss / get_local_loss / __call__
 *     udt_wgrad_bf16, udt_colsum_bf16                     weight / bias gradients of the trained nn.Linear layers
 *     udt_adamw_f32                                       torch.optim.AdamW.step (diffusion.py:49-51,202-222)
 *     (backward-data of linears and convolutions: udt_gemm on re-packed weights)
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; kernels borrow it for the stream-ordered
 *     duration of the launch; the library allocates nothing on the data path and keeps no pointer.
 *   - activations are bf16, channel-last ("NHWC": [batch, pixels, channels]); weights are bf16,
 *     [out_features, K] with K contiguous; statistics / sampler state are fp32.
 *   - return value: 0 on success, a negative udt_status otherwise.  Nothing throws or aborts.
 *   - all launches are asynchronous on `stream` (a hipStream_t passed as void*).
 */
#ifndef UDT_KERNELS_H
#define UDT_KERNELS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  UDT_OK = 0,
  UDT_ERR_BAD_SHAPE = -1,     /* dimension not supported by the kernel (alignment / range)      */
  UDT_ERR_BAD_ARG = -2,       /* null pointer / inconsistent flags                                */
  UDT_ERR_WORKSPACE = -3,     /* workspace too small (see udt_gemm_workspace_bytes)               */
  UDT_ERR_HIP = -4,           /* a HIP runtime call failed; see udt_last_hip_error                */
  UDT_ERR_NO_DEVICE = -5,     /* no gfx950 device visible                                         */
  UDT_ERR_ASYNC = -6          /* a stream-K launch timed out waiting for a partner workgroup (its output
                                 tile is NaN-poisoned); reported by udt_check_async_error                */
} udt_status;

/* element types of packed weights (udt_pack_*) */
typedef enum {
  UDT_DTYPE_F32 = 0,
  UDT_DTYPE_BF16 = 1,
  UDT_DTYPE_FP8_E4M3 = 2      /* OCP e4m3 (not the fnuz format of MI300X) + one fp32 scale per output channel */
} udt_dtype;

/* ---- epilogue / mode flags for udt_gemm ------------------------------------------------------- */
#define UDT_GEMM_OUT_F32     (1 << 0)  /* out is fp32 instead of bf16                              */
#define UDT_GEMM_GEGLU       (1 << 1)  /* out[:, j] = x_j * gelu_erf(gate_j); weight rows packed
                                          in [32 x | 32 gate] blocks (see pack_geglu in Python)   */
#define UDT_GEMM_RELU        (1 << 2)
#define UDT_GEMM_TRANSPOSED  (1 << 3)  /* out is [batch_of_row][N][rows_per_batch] (V^T for attn)  */
#define UDT_GEMM_CONV        (1 << 4)  /* implicit-GEMM convolution, A gathered from NHWC sources   */
#define UDT_GEMM_SILU_OUT    (1 << 5)  /* out = silu(acc...) (time_embed MLP)                      */
/* (1 << 6) was UDT_GEMM_FP8 — round 2's first-generation fp8 path (unit block scales, a per-tensor activation scale written by a
   quantising LayerNorm kernel): slower than the LayerNorm-folded bf16 GEMMs, retired in round 5 in favour of UDT_GEMM_MX8. */
#define UDT_GEMM_MX8         (1 << 7)  /* BASELINE config #5: a is an MX8 activation — OCP e4m3
                                          bytes [M, lda] + one E8M0 block scale per 32 K-elements of a
                                          row (a_scale) — and w is e4m3 [N, ldw] + per-channel fp32 scales
                                          (colscale); K % 128 == 0, lda / ldw % 16 == 0.  Plain, LayerNorm-
                                          folded (rowstat_in) and GEGLU epilogues on the lean kernel family;
                                          see the q8_out / rowstat_* fields.                              */

typedef struct {
  /* operands */
  const void* a;        /* bf16 [M, lda] (plain) or NHWC source 1 [B, Hin, Win, C1] (conv)        */
  const void* a2;       /* conv only: optional NHWC source 2 [B, Hin, Win, C2] (channel concat)   */
  const void* w;        /* bf16 [N, K] K-contiguous; conv: k = (ky*ksize + kx)*(C1+C2) + c        */
  const float* bias;    /* [N] fp32 or NULL                                                       */
  const void* residual; /* bf16 [M, ldr] or NULL, added in the epilogue                           */
  const float* rowvec;  /* fp32 [M / rows_per_batch, N] or NULL (time-embedding broadcast add)    */
  void* out;            /* bf16 / fp32 [M, ldo] (or transposed, see flag)                         */
  int32_t M, N, K;
  int32_t lda, ldo, ldr;
  int32_t ldw;          /* elements between rows of w (0 = K)                                       */
  /* batched GEMM (grid.z): element strides between problems; batch = 1 for none                  */
  int32_t batch;
  int64_t stride_a, stride_w, stride_out, stride_res;
  /* conv geometry (UDT_GEMM_CONV)                                                                */
  int32_t Hin, Win, C1, C2, Hout, Wout;
  int32_t ksize;        /* 1 or 3                                                                 */
  int32_t stride;       /* 1 or 2                                                                 */
  int32_t pad_t, pad_l; /* zero padding before the first row / column (bottom/right implicit)     */
  int32_t upsample;     /* 1: sources are read through a nearest x2 upsample                      */
  int32_t rows_per_batch; /* rows of M belonging to one sample (rowvec / transposed addressing)   */
  int32_t ld_rowvec;    /* elements between rows of rowvec (0 = N)                                 */
  int32_t flags;
  float alpha;          /* acc * alpha before bias (softmax scale for QK^T GEMMs); 1.0 default    */
  /* fused GroupNorm (north star: "3x3 conv + GroupNorm + SiLU fused blocks"; reference chain
     openaimodel.py:183-187,218-231, util.py:258-275, model.py:128-148)                                             */
  const float* colscale;/* UDT_GEMM_MX8 only: fp32 [N] per-output-channel weight scales (in packed row order), or NULL */
  const float* in_scsh; /* 3x3 / stride 1 / pad 1 convolutions only (see udt_gn_silu_conv3x3_fwd): per-(sample,
                           channel) scale / shift from udt_gn_finalize, applied — with in_act — to the input patch
                           as it is staged in LDS: y = act(x * scale + shift); zero padding stays zero.  NULL = off */
  int32_t in_act;       /* 0 none, 1 SiLU                                                                          */
  float* colstats;      /* optional output: per-(row slot, output column) partial (sum, sum of squares) of the
                           result, fp32 [udt_gemm_colstats_slots][N][2]: the GroupNorm statistics of the NEXT
                           layer come out of this layer's epilogue.  NULL = off                                   */
  int32_t cu_share;     /* number of launch streams that share the device with this call (0 / 1: none).
                           The persistent stream-K kernels wait on partner workgroups, so all their
                           workgroups must be resident: the launch is planned for 1/cu_share of the CUs.
                           udt_gemm_workspace_bytes must be asked with the same value.                  */
  /* LayerNorm prologue (udt_ln_gemm_fwd; reference attention.py:310-339 `attn1(norm1(x))` / `ff(norm3(x))`): `a` holds
     the RAW rows x, `w` = gamma o W (LayerNorm scale folded into the weight columns), `bias` = c_n = sum_k beta_k W_nk
     + linear bias, ln_colsum = s_n = sum_k w_nk (fp32 [N], of the bf16-rounded folded weights); the kernel takes every
     row's mean / rstd from the A tiles it streams and computes out = rstd (x w^T - mean s) + c.  K = normalised width. */
  const float* ln_colsum;
  float ln_eps;
  /* MX8 activations (UDT_GEMM_MX8 and the emitting epilogues; reference call sites: every nn.Linear of the transformer blocks,
     attention.py:44-70,193-199,375-411).  An MX8 activation of C columns is (elements: e4m3 bytes [M, ld >= C];
     scales: uint32 [ceil(C / 128)][M], byte j of dword (t, m) = E8M0 scale of columns [128 t + 32 j, + 32) of row m:
     value = e4m3 * 2^(scale - 127)).                                                                                   */
  const void* a_scale;     /* UDT_GEMM_MX8: the block scales of `a`                                                     */
  void* q8_out;            /* optional second output: the result (after bias / residual / GEGLU, as rounded for `out`) again
                              as an MX8 activation for the next GEMM; `out` may then be NULL.  Lean
                              128 x 128 plans and the row-resident LayerNorm-folded K = 320 plan only (udt_gemm_q8_ok), N % 32 == 0
                              (GEGLU: N % 64 == 0)                                                                         */
  void* q8_scale;          /* its block scales (see above)                                                               */
  int32_t ld_q8;           /* bytes between rows of q8_out (% 8 == 0)                                                    */
  int32_t q8_fixed_col;    /* 0 = off; else result columns >= q8_fixed_col (a multiple of 32; 64 on the row-resident plan) are written as e4m3(value * q8_fixed_mul > 0),
                              clamped to +-448, with the unit scale byte 127 instead of block scales: the v third of a q|k|v
                              projection feeding udt_attn_mx8_fwd, whose P V product contracts over keys                        */
  float q8_fixed_mul;
  float* rowstat_out;      /* optional, with q8_out (not GEGLU): fp32 [udt_gemm_rowstat_parts(d)][M][2] partial (sum, sum of
                              squares) of every result row — the LayerNorm statistics of a LayerNorm-folded MX8 consumer  */
  const float* rowstat_in; /* UDT_GEMM_MX8 with ln_colsum: the partial row statistics of `a` its producer emitted,
                              fp32 [rowstat_in_parts][M][2]; summed in index order (deterministic)                        */
  int32_t rowstat_in_parts;
} udt_gemm_desc;

/* workspace (bytes) udt_gemm needs for this problem (split-K slabs); 0 if none.  The first 4 KiB of a workspace are
 * the kernels' slab flags + error word: zero them once when the buffer is allocated (they are zero again after
 * every successful launch) and give concurrent streams separate workspaces. */
size_t udt_gemm_workspace_bytes(const udt_gemm_desc* d);
/* Limits: the MFMA kernels address one operand (one batch element) through 31-bit byte offsets — an A or W operand of 2 GiB or more
 * is only served for outputs of <= 64 columns (the first-generation 256 x 64 kernel); wider problems of that size return
 * UDT_ERR_BAD_SHAPE (udt_gemm_workspace_bytes answers 0 for them: nothing to plan).  No shape of the UNet / VAE comes near. */
int udt_gemm(const udt_gemm_desc* d, void* workspace, size_t workspace_bytes, void* stream);
/* Column-statistics geometry of udt_gemm for this problem: rows of the output covered by one slot of `colstats`
 * (32 or 64; slot s covers rows [s*rows, (s+1)*rows) of M for plain GEMMs / gathered convolutions and the same number of
 * pixels of ONE image in tile order for the patch-staged convolution — either way the slots of a sample are contiguous:
 * slots_per_sample = rows_per_batch / rows) and the number of slots.  0 = this problem cannot emit them (transposed /
 * GEGLU / fp32 outputs, first-generation kernel, rows_per_batch not a multiple of the slot). */
int32_t udt_gemm_colstats_rows(const udt_gemm_desc* d);
/* Parts of the partial row statistics (rowstat_out) udt_gemm would emit for this problem = N / (columns per wave of the plan);
 * 0 = the plan has no MX8-emitting epilogue (q8_out / rowstat_out must then be NULL). */
int32_t udt_gemm_rowstat_parts(const udt_gemm_desc* d);
/* 1 if udt_gemm would take this descriptor WITH its q8_out / q8_scale (an MX8-emitting epilogue exists for the plan), else 0. */
int32_t udt_gemm_q8_ok(const udt_gemm_desc* d);
int32_t udt_gemm_colstats_slots(const udt_gemm_desc* d);
/* 1 if udt_gemm accepts `in_scsh` for this problem (patch-staged 3x3 convolution geometry; one or two NHWC sources). */
int32_t udt_gemm_in_scsh_ok(const udt_gemm_desc* d);
/* The north star's fused block under its own name: GroupNorm(+SiLU) applied on the staged input patch -> 3x3 conv ->
 * epilogue (bias, time-embedding row vector, residual) -> optional statistics of the output.  Same as udt_gemm with a
 * descriptor that has UDT_GEMM_CONV, ksize 3, stride 1, pad 1 and in_scsh set; anything else is UDT_ERR_BAD_ARG. */
int udt_gn_silu_conv3x3_fwd(const udt_gemm_desc* d, void* workspace, size_t workspace_bytes, void* stream);
/* LayerNorm -> linear in one launch (SURVEY.md §8b `udt_ln_gemm_fwd`; reference attention.py:310-339: `attn1(norm1(x))`,
 * `ff(norm3(x))`, LayerNorm over the last dimension = the GEMM's K).  Descriptor as for udt_gemm with the ln_colsum /
 * ln_eps fields set (see udt_gemm_desc); plain or GEGLU epilogue, optional residual.  UDT_ERR_BAD_SHAPE when the
 * problem is outside the lean GEMM family (N <= 64, fp32 / transposed outputs, batched). */
int udt_ln_gemm_fwd(const udt_gemm_desc* d, void* workspace, size_t workspace_bytes, void* stream);
/* Synchronises `stream` and reports (UDT_ERR_ASYNC) whether any udt_gemm launch that used `workspace` gave up waiting
 * for a partner workgroup since the last check; in that case the header is re-zeroed so the workspace stays usable.
 * Callers check at their natural sync points (the sampler: once per sampling loop). */
int udt_check_async_error(void* workspace, size_t workspace_bytes, void* stream);

/* ---- attention -------------------------------------------------------------------------------- */
/* Flash attention forward, head_dim 64, no mask:  O = softmax(Q K^T * scale) V
 *   q  : bf16, row (b, tok) at q  + b*q_bstride  + tok*ldq + h*64
 *   k  : bf16, row (b, tok) at k  + b*k_bstride  + tok*ldk + h*64
 *   vt : bf16 V transposed: element (b, h*64+d, tok) at vt + b*vt_bstride + (h*64+d)*ldvt + tok
 *   o  : bf16, row (b, tok) at o  + b*o_bstride  + tok*ldo + h*64
 * nk must be a multiple of 8. */
int udt_attn_fwd(const void* q, const void* k, const void* vt, void* o,
                 int32_t batch, int32_t heads, int32_t nq, int32_t nk,
                 int32_t ldq, int32_t ldk, int32_t ldvt, int32_t ldo,
                 int64_t q_bstride, int64_t k_bstride, int64_t vt_bstride, int64_t o_bstride,
                 float scale, void* stream);

/* Short-context attention (context length L <= 16; text cross-attention, LabelEncoder self-attn):
 *   q  : bf16 [batch, nq, heads*head_dim] (ldq), k/v : bf16 rows (b, l) at k + (b*L + l)*ldkv + h*head_dim
 *   o  : bf16 [batch, nq, heads*head_dim] (ldo)
 *   probs (optional, may be NULL): fp32 [batch*heads, nq, L] — the softmax probabilities the
 *   reference caches in attn_map_cache (attention.py:165-169), "(b h) n l" order.
 * head_dim must be a multiple of 64. */
int udt_xattn_fwd(const void* q, const void* k, const void* v, void* o, float* probs,
                  int32_t batch, int32_t heads, int32_t head_dim, int32_t nq, int32_t L,
                  int32_t ldq, int32_t ldkv, int32_t ldo, float scale, void* stream);

/* The same with V row-major like K (row (b, tok) at v + b*v_bstride + tok*ldv + h*64): q, k and v may then be column
 * ranges of ONE q|k|v projection output (reference attention.py:193-199: to_q / to_k / to_v on the same input). */
int udt_attn_rowv_fwd(const void* q, const void* k, const void* v, void* o,
                      int32_t batch, int32_t heads, int32_t nq, int32_t nk,
                      int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                      int64_t q_bstride, int64_t k_bstride, int64_t v_bstride, int64_t o_bstride,
                      float scale, void* stream);
/* The same launch that ALSO writes O as an MX8 activation (udt_gemm_desc "MX8 activations": q8_out e4m3 [batch * nq, ld_q8], column
 * h * 64 + d; q8_scale uint32 [heads * 64 / 128][batch * nq]) for the e4m3 `to_out` GEMM of BASELINE config #5 (reference
 * attention.py:246-262).  heads even, ld_q8 >= heads * 64, ld_q8 % 4 == 0, o_bstride == nq * ldo. */
int udt_attn_rowv_q8_fwd(const void* q, const void* k, const void* v, void* o,
                         int32_t batch, int32_t heads, int32_t nq, int32_t nk,
                         int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                         int64_t q_bstride, int64_t k_bstride, int64_t v_bstride, int64_t o_bstride,
                         float scale, void* q8_out, void* q8_scale, int32_t ld_q8, void* stream);

/* BASELINE config #5's "fp8 attention": the same self-attention on e4m3 operands (v_mfma_scale_f32_32x32x64_f8f6f4 for Q K^T and
 * P V; fp32 softmax).  q, k, v are column ranges of ONE MX8 activation qkv8 [batch * n, ld8] — q at column h * 64, k at C + h * 64,
 * v at 2 C + h * 64, C = heads * 64 — written by the q|k|v projection's emitting epilogue (udt_gemm_desc.q8_out) with
 * q8_fixed_col = 2 C: q and k carry block scales along the head dimension (qkv_scale, uint32 [ld8 / 128 rounded up][batch * n]),
 * v was multiplied by q8_fixed_mul; v_inv = 1 / q8_fixed_mul.  o: bf16 rows b * n + i of ldo elements; q8_out / q8_scale / ld_q8
 * (optional): O again as an MX8 activation, as udt_attn_rowv_q8_fwd (heads even for it).  n % 4 == 0, ld8 % 16 == 0, qkv8 16-byte aligned.
 * Reference: sgm/modules/attention.py:236-248. */
int udt_attn_mx8_fwd(const void* qkv8, const void* qkv_scale, void* o, int32_t batch, int32_t heads, int32_t n,
                     int32_t ld8, int32_t ldo, float scale, float v_inv, void* q8_out, void* q8_scale, int32_t ld_q8,
                     void* stream);

/* Flash attention forward for ONE head of 512 dims: the AutoencoderKL mid-block attention (reference
 * sgm/modules/diffusionmodules/model.py:236-260, MemoryEfficientAttnBlock.attention: xformers.ops.memory_efficient_attention on
 * [B, H*W, 512]; AttnBlock.attention, model.py:169-190, is the same product through softmax(q k^T / sqrt(C)) v).
 *   q, k, v : bf16 rows (b, tok) at x + b*x_bstride + tok*ldx, 512 dims each — they may be column ranges of ONE q|k|v projection
 *   o       : bf16 rows of 512 dims (ldo); nq / nk arbitrary (keys past nk are masked), 16-byte aligned q / k / v, ld % 8 == 0.
 * No [nq, nk] score tensor is materialised (the score tile of 32 queries x 32 keys lives in registers / LDS). */
int udt_attn512_fwd(const void* q, const void* k, const void* v, void* o,
                    int32_t batch, int32_t nq, int32_t nk,
                    int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                    int64_t q_bstride, int64_t k_bstride, int64_t v_bstride, int64_t o_bstride,
                    float scale, void* stream);

/* The same product with the KEYS split over workgroups (flash-decoding form) for grids whose query tiles alone leave most of
 * the 256 CUs idle — a single 512 x 512 image is 64 query tiles, each walking all 4096 keys: the keys are cut into up to 8
 * slices of >= 256 keys, every (query tile, slice) workgroup parks its unnormalised fp32 partial O and (running maximum, row
 * sum) in `workspace`, and a merge launch on the same stream combines the slices in index order (deterministic).
 *   udt_attn512_workspace_bytes : bytes the split form needs for (batch, nq, nk); 0 = the grid is large enough, no split.
 *   udt_attn512_split_fwd       : as udt_attn512_fwd; workspace 16-byte aligned, >= the bytes above (UDT_ERR_WORKSPACE
 *                                 otherwise); with workspace == NULL, or when no split is planned, it IS udt_attn512_fwd. */
size_t udt_attn512_workspace_bytes(int32_t batch, int32_t nq, int32_t nk);
int udt_attn512_split_fwd(const void* q, const void* k, const void* v, void* o,
                          int32_t batch, int32_t nq, int32_t nk,
                          int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                          int64_t q_bstride, int64_t k_bstride, int64_t v_bstride, int64_t o_bstride,
                          float scale, void* workspace, size_t workspace_bytes, void* stream);

/* Masked small attention (the OCR scorer's decoder: nn.MultiheadAttention of PARSeq's DecoderLayer, reference
 * src/parseq/strhub/models/parseq/modules.py:35-36,57-70 -> torch's scaled-dot-product with attn_mask and
 * key_padding_mask).  Few queries against a short key set, any head_dim that is a multiple of 8 up to 64:
 *   q : bf16 rows (b, i) at q + b*q_bstride + i*ldq + h*head_dim      i < nq
 *   k, v : bf16 rows (b, l) at k + b*k_bstride + l*ldk + h*head_dim   l < lk, lk <= 256, lk*head_dim <= 8192
 *   o : bf16 rows (b, i) at o + b*o_bstride + i*ldo + h*head_dim
 *   mask (optional): fp32 [nq, ldmask] added to the scaled scores (-inf = masked), shared by batch and heads
 *   kpm  (optional): uint8 [batch, lk], non-zero = the key is padding (ignored)
 * fp32 scores, softmax and accumulation.  A query whose keys are all masked yields zeros (torch yields NaN). */
int udt_mattn_fwd(const void* q, const void* k, const void* v, void* o, const float* mask, const uint8_t* kpm,
                  int32_t batch, int32_t heads, int32_t head_dim, int32_t nq, int32_t lk,
                  int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo, int32_t ldmask,
                  int64_t q_bstride, int64_t k_bstride, int64_t v_bstride, int64_t o_bstride, float scale, void* stream);

/* The text cross-attention branch of a transformer block as ONE kernel (csrc/tattn.hip):
 *   out = x + to_out(softmax(to_q(LayerNorm(x)) K^T * scale) V) + bias        reference sgm/modules/attention.py:140-174,326-333
 * The context is constant during sampling and has L <= 12 tokens, so udt_tattn_prepare folds it ONCE per batch into
 * per-sample tables, opaque to the caller (exact re-association; stored in the MFMA fragment order of the consuming kernel): A' [B][hp][C] bf16 (LayerNorm gamma, to_q and K folded; hp = udt_tattn_hp(heads)
 * = 16 columns per head, rounded up to 32), sc [B][hp][2] fp32 (the LayerNorm mean / beta terms), BmT [B][C][hp] bf16 (V and
 * to_out folded).  udt_tattn_fused then needs only the raw rows: x, out bf16 [B * n_tok, C]; the kernel is
 * instantiated for the UNet's three widths — C = 64 * heads with heads = 5, 10, 20 — anything else is UDT_ERR_BAD_SHAPE
 * (n_tok % 64 == 0, or % 32 for C = 1280); the first `zero_samples` samples attend to an all-zero context and get
 * x + bias (no tables needed when zero_samples == B).
 *   kv: bf16 [B, L, ldkv], k in columns [0, C), v in [C, 2C) (the hoisted to_k|to_v projection); wq: bf16 [C, ldwq] (to_q
 *   weight, rows = output features), wo: bf16 [C, ldwo] (to_out weight); gamma / beta: t_norm; bias: to_out bias. */
int32_t udt_tattn_hp(int32_t heads);
int udt_tattn_prepare(const void* kv, int32_t ldkv, const void* wq, int32_t ldwq, const void* wo, int32_t ldwo,
                      const float* gamma, const float* beta, void* A, float* sc, void* BmT, int32_t B, int32_t L,
                      int32_t C, int32_t heads, float scale, void* stream);
int udt_tattn_fused(const void* x, void* out, const void* A, const float* sc, const void* BmT, const float* bias,
                    int32_t B, int32_t n_tok, int32_t C, int32_t heads, int32_t zero_samples, float eps, void* stream);
/* The same launch that ALSO writes its result as an MX8 activation (udt_gemm_desc "MX8 activations": q8_out e4m3 [B * n_tok, C],
 * q8_scale uint32 [C / 128][B * n_tok]) and the partial row statistics (rowstat_out fp32 [udt_tattn_rowstat_parts][B * n_tok][2]) for
 * the LayerNorm-folded GEGLU projection that consumes it on the e4m3 path (BASELINE config #5; reference attention.py:326-339:
 * `x = t_attn(...) + x; x = ff(norm3(x)) + x`).  C = 640 / 1280.  udt_tattn_rowstat_parts: 0 = no such instance for the shape. */
int32_t udt_tattn_rowstat_parts(int32_t B, int32_t n_tok, int32_t C);
int udt_tattn_fused_q8(const void* x, void* out, const void* A, const float* sc, const void* BmT, const float* bias,
                       int32_t B, int32_t n_tok, int32_t C, int32_t heads, int32_t zero_samples, float eps, void* q8_out,
                       void* q8_scale, float* rowstat_out, void* stream);

/* Row softmax of a bf16 [rows, cols] matrix in place (ld elements between rows), fp32 math. */
int udt_softmax_rows(void* x, int64_t rows, int32_t cols, int32_t ld, void* stream);

/* ---- normalisation ---------------------------------------------------------------------------- */
/* GroupNorm statistics over NHWC bf16 x [B, HW, C] with G groups (C % G == 0, C % 8 == 0).
 * Writes per-chunk partial (sum, sumsq) to `partials` (fp32 [B, nchunks, G, 2]); nchunks is returned
 * by udt_gn_nchunks(HW, C). */
int32_t udt_gn_nchunks(int64_t HW, int32_t C);
int udt_gn_stats(const void* x, const void* x2, float* partials, int32_t B, int64_t HW, int32_t C, int32_t C2,
                 int32_t G, void* stream);
/* y = act((x - mean) * rstd * gamma + beta); act: 0 none, 1 SiLU.  y may alias x when x2 == NULL.
 * x2 (optional, C2 channels) is a second NHWC source concatenated after x's C channels (UNet skip concat,
 * openaimodel.py:620); statistics and output cover the C + C2 concatenated channels, y is [B, HW, C + C2]. */
int udt_gn_apply(const void* x, const void* x2, void* y, const float* partials, const float* gamma,
                 const float* beta, int32_t B, int64_t HW, int32_t C, int32_t C2, int32_t G, float eps, int32_t act,
                 void* stream);
/* One-launch GroupNorm (+ optional SiLU) for shapes whose per-sample group strips fit the caches (udt_gn_strip_ok): a
 * workgroup owns a few consecutive groups of one sample, reads them once for the statistics and again (out of L2) to
 * normalise.  Same arguments and result as udt_gn_stats + udt_gn_apply (statistics summed in a different order). */
int32_t udt_gn_strip_ok(int32_t B, int64_t HW, int32_t C1, int32_t C2, int32_t G);
int udt_gn_strip(const void* x, const void* x2, void* y, const float* gamma, const float* beta, int32_t B, int64_t HW,
                 int32_t C1, int32_t C2, int32_t G, float eps, int32_t act, void* stream);

/* GroupNorm statistics from producer epilogues -> per-(sample, channel) scale / shift for udt_gemm's in_scsh.
 *   stats1 fp32 [B * slots1][C1][2] (+ optional stats2 [B * slots2][C2][2] for a channel concat x ‖ x2): the colstats of
 *   the layer(s) that produced the input; slotsN = slots per sample.  Groups run over the C1 + C2 concatenated channels.
 *   scsh out fp32 [B][(C1+C2)/64][2][64]: chunk-blocked (scale of 64 channels, then their shift) — the layout the
 *   convolution's LDS-DMA reads.  scale = rstd * gamma, shift = beta - mean * scale; fp32 partials, fp64 combine.
 *   (C1 + C2) % 64 == 0, (C1 + C2) % G == 0. */
int udt_gn_finalize(const float* stats1, int32_t slots1, int32_t C1, const float* stats2, int32_t slots2, int32_t C2,
                    const float* gamma, const float* beta, float* scsh, int32_t B, int64_t HW, int32_t G, float eps,
                    void* stream);

/* GroupNorm (+ SiLU) from a finished scale / shift table: y = act(x * scale[b, c] + shift[b, c]) with `scsh` as written by
 * udt_gn_finalize from the column statistics the PRODUCERS of x (and x2) emitted in their epilogues (udt_gemm_desc.colstats) —
 * the statistics pass over the tensor is gone, this is the one remaining read + write of reference GroupNorm32 -> SiLU
 * (sgm/modules/diffusionmodules/util.py:214-216, openaimodel.py:183-187).  Two sources = the channel concat of a decoder
 * ResBlock's input; C1 + C2 a multiple of 64. */
int udt_gn_apply_scsh(const void* x, const void* x2, void* y, const float* scsh, int32_t B, int64_t HW, int32_t C1,
                      int32_t C2, int32_t act, void* stream);

/* The one-launch strip GroupNorm (udt_gn_strip) when the producers of x (and x2) emitted column statistics
 * (udt_gemm_desc.colstats: fp32 [B * slots][C][2]): the statistics pass over the strip becomes a sum over `slots` records per
 * channel, the launch is the normalise(+SiLU) pass alone.  Same shapes as udt_gn_strip (udt_gn_strip_ok). */
int udt_gn_strip_stats(const void* x, const void* x2, void* y, const float* stats1, int32_t slots1, const float* stats2,
                       int32_t slots2, const float* gamma, const float* beta, int32_t B, int64_t HW, int32_t C1, int32_t C2,
                       int32_t G, float eps, int32_t act, void* stream);

/* LayerNorm over the last dim of bf16 [rows, C] (C % 8 == 0, C <= 4096). */
int udt_layernorm(const void* x, void* y, const float* gamma, const float* beta,
                  int64_t rows, int32_t C, float eps, void* stream);
/* ---- sampler / boundary elementwise ------------------------------------------------------------ */
/* UNet input for one CFG step: x fp32 NCHW [B,4,h,w] -> xin bf16 NHWC [2B, h*w, cpad]; channels 0..3 of
 * both halves = x * c_in; the other channels (mask / masked latent / zero pad) are left untouched. */
int udt_unet_input(const float* x, void* xin, int32_t B, int32_t hw, int32_t cpad, float c_in, void* stream);
/* eps fp32 [2B, hw, ld_eps] (uncond half first) -> x fp32 NCHW [B,4,h,w] updated in place:
 *   den_u = x + c_out*eps_u ; den_c = x + c_out*eps_c (c_out = -quantised sigma) ; den = den_u + scale*(den_c - den_u)
 *   d = (x - den)/sigma ; x += d*(sigma_next - sigma)          (optionally writes den) */
int udt_cfg_euler_step(float* x, const float* eps, float* denoised_out, int32_t B, int32_t hw, int32_t ld_eps,
                       float c_out, float sigma, float sigma_next, float cfg_scale, void* stream);
/* z = scale * (mean + exp(0.5*clamp(logvar,-30,20)) * noise); moments fp32 [B, hw, ldm] NHWC (mean ch 0..3,
 * logvar ch 4..7), noise fp32 NCHW [B,4,h,w], z fp32 NCHW [B,4,h,w]. */
int udt_posterior_sample(const float* moments, const float* noise, float* z, int32_t B, int32_t hw, int32_t ldm,
                         float scale, void* stream);
/* fp32 NCHW [B,C,HW] -> bf16 NHWC [B,HW,cpad] (channels >= C zero-filled), value * scale */
int udt_nchw_to_nhwc(const float* x, void* y, int32_t B, int32_t C, int64_t HW, int32_t cpad, float scale,
                     void* stream);
/* bf16 or fp32 NHWC [B,HW,ld] (first C channels) -> fp32 NCHW [B,C,HW] */
int udt_nhwc_to_nchw(const void* x, float* y, int32_t B, int32_t C, int64_t HW, int32_t ld, int32_t src_is_f32,
                     void* stream);
/* write fp32 NCHW [B,C,HW] source into channels [c0, c0+C) of a bf16 NHWC buffer [B,HW,cpad] */
int udt_nhwc_set_channels(const float* src, void* dst, int32_t B, int32_t C, int64_t HW, int32_t cpad, int32_t c0,
                          void* stream);
/* out bf16 [n_tok, D] = table[idx[i]] + pe[i % L]  (table fp32 [V, D], pe fp32 [L, D], idx int32) */
int udt_embed_tokens(const int32_t* idx, const float* table, const float* pe, void* out, int32_t n_tok, int32_t L,
                     int32_t D, void* stream);
/* out bf16 [n, dim]: [cos(t f_k) | sin(t f_k)], f_k = exp(-ln(10000) k / (dim/2)); t int64-valued fp32 */
int udt_timestep_embedding(const float* t, void* out, int32_t n, int32_t dim, void* stream);
/* bilinear x1/8 (align_corners False) of fp32 [B,1,H,W] -> fp32 [B,1,H/8,W/8]: mean of the centre 2x2 */
int udt_mask_downsample(const float* mask, float* out, int32_t B, int32_t H, int32_t W, void* stream);
/* per-layer local-loss term: probs fp32 [B*heads, n, L] (n = size*size), mask fp32 [B,1,Hm,Wm],
 * seg_mask fp32 [B, seg_l] -> loss fp32 [B] (+= accumulate): -min_l( max_n(mask_n * blur3x3(mean_h p)) + 1 - seg ) */
int udt_local_loss(const float* probs, const float* mask, const float* seg_mask, const float* gkernel9,
                   float* loss_accum, int32_t B, int32_t heads, int32_t size, int32_t L, int32_t seg_l,
                   int32_t Hm, int32_t Wm, void* stream);
/* the same for n_samples = k * mask_batch maps scored against mask / seg row (sample % mask_batch): the candidates of the noise
 * search (reference sampling.py get_init_noise, loss.py:192-235) as extra batch entries of one launch; loss_accum fp32 [n_samples] */
int udt_local_loss_tiled(const float* probs, const float* mask, const float* seg_mask, const float* gkernel9,
                         float* loss_accum, int32_t n_samples, int32_t mask_batch, int32_t heads, int32_t size, int32_t L,
                         int32_t seg_l, int32_t Hm, int32_t Wm, void* stream);
/* ---- backward (dX only) for attend-and-excite, SURVEY 8f-4 — reference sampling.py:233-252: torch.autograd.grad(local_loss, x)
 * through the UNet; csrc/backward.hip.  Linears / convolutions have no entry of their own: their backward-data is udt_gemm on
 * re-packed weights (W^T; 180-degree rotated taps with the channel roles swapped). ------------------------------------------------- */
/* flash-attention backward, head_dim 64 (reference attention.py:236-248 xformers memory_efficient_attention under autograd):
 * q, k, v: column ranges of one [batch * n, ldq] bf16 matrix (head h at columns h * 64 of each pointer), o / d_o [batch * n, ldo];
 * dq, dk, dv: column ranges of one [batch * n, ldd] bf16 matrix.  lse_ws / dsum_ws: fp32 [batch * heads * n] scratch each (the
 * log-sum-exp of every score row and rowsum(dO o O), written by the first launch, read by the second).  Two launches, no atomics. */
int udt_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, void* dq, void* dk, void* dv,
                 float* lse_ws, float* dsum_ws, int32_t batch, int32_t heads, int32_t n, int32_t ldq, int32_t ldo, int32_t ldd,
                 float scale, void* stream);
/* text cross-attention backward (reference attention.py:140-175 CrossAttention under autograd; context constant): probs fp32
 * [batch * heads, nq, L] as udt_xattn_fwd wrote them, d_probs (optional) the loss gradient with respect to them, d_o (optional) bf16
 * [batch * nq, ldo] the gradient of the attention output -> dq bf16 [batch * nq, lddq] (head h at columns h * 64); L <= 16, and
 * the sigmoid branch for L == 1. */
int udt_xattn_bwd(const void* k, const void* v, const float* probs, const float* d_probs, const void* d_o, void* dq, int32_t batch,
                  int32_t heads, int32_t head_dim, int32_t nq, int32_t L, int32_t ldkv, int32_t ldo, int32_t lddq, float scale,
                  void* stream);
/* gradient of udt_local_loss_tiled's per-layer term with respect to probs, times `weight` (1 / number of contributing layers),
 * ADDED into d_probs fp32 [n_samples * heads, n, L] (zero-initialised by the caller); loss_accum (optional) += the term itself
 * (reference loss.py:192-235 under autograd: arg-min token, arg-max pixel, the 3x3 blur stencil, the head mean) */
int udt_local_loss_bwd(const float* probs, const float* mask, const float* seg_mask, const float* gkernel9, float* d_probs,
                       float* loss_accum, float* scratch /* n_samples * seg_l * 2 floats: the tokens' scores */, int32_t n_samples,
                       int32_t mask_batch, int32_t heads, int32_t size, int32_t L, int32_t seg_l, int32_t Hm, int32_t Wm, float weight,
                       void* stream);
/* LayerNorm backward-data (nn.LayerNorm of attention.py:310-339): x, dy bf16 [rows, C] -> dx bf16 (+ add bf16 [rows, C] if given:
 * the gradient that arrives over the residual connection); statistics recomputed from x; C % 8 == 0, C <= 2048 */
int udt_layernorm_bwd(const void* x, const void* dy, const float* gamma, const void* add, void* dx, int64_t rows, int32_t C, float eps,
                      void* stream);
/* GroupNorm (+ SiLU when silu != 0) backward-data (GroupNorm32 / nn.SiLU of openaimodel.py:183-187, attention.py:375): x, dy bf16
 * NHWC [B, HW, C] -> dx (+ add); statistics recomputed from x; (C / groups) even.  partials: fp32 scratch of
 * 2 * B * udt_gn_nchunks(HW, C) * groups * 2 values — with it (and C % 8 == 0, 256 % groups == 0) the gradient runs as three launches
 * at the forward GroupNorm's parallelism (chunk partials of sum x / sum x^2, of sum t / sum t xhat, then the result); NULL: one
 * workgroup per (sample, group) */
int udt_gn_bwd(const void* x, const void* dy, const float* gamma, const float* beta, const void* add, void* dx, float* partials, int32_t B,
               int32_t HW, int32_t C, int32_t groups, float eps, int32_t silu, void* stream);
/* GEGLU (attention.py:44-52) on STORED pre-activations ag bf16 [rows, 2 * inner] = [x | gate]: out = x * gelu(gate) [rows, inner];
 * backward: dag [rows, 2 * inner] from dy [rows, inner] (the inference path's fused GEMM epilogue keeps no pre-activations) */
int udt_geglu_fwd(const void* ag, void* out, int64_t rows, int32_t inner, void* stream);
int udt_geglu_bwd(const void* ag, const void* dy, void* dag, int64_t rows, int32_t inner, void* stream);
/* nearest x2 upsampling backward (openaimodel.py:99-101 under autograd): dy bf16 NHWC [B, 2H, 2W, C] -> dx [B, H, W, C] = 2 x 2 sums */
int udt_sum2x2_bf16(const void* dy, void* dx, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
/* context tokens minus their per-sample mean over the L tokens, as bf16: x fp32 [B, L, D] -> out bf16 [B, L, D].  The softmax over the
 * tokens (attention.py:155-160) and its backward are invariant under that shift of the keys / values; used by the attend-and-excite
 * tape so that the bf16 k|v projection keeps the DIFFERENCES between the (nearly equal) label-embedding tokens */
int udt_center_tokens(const float* x, void* out, int32_t B, int32_t L, int32_t D, void* stream);
/* ---- training step of the text cross-attention (second half; reference diffusion.py:138-172,202-222, loss.py:131-176,
 * 237-286; configs/train/textdesign_sd_2.yaml:4-6: only t_attn / t_norm are trained).  dW = dY^T X runs as udt_gemm on operands
 * transposed by udt_transpose_bf16. ---------------------------------------------------------------------------------------------- */
/* in bf16 [R, ld] (first C columns) -> out bf16 [C, Rp], Rp >= R a multiple of 64 (columns >= R zero) */
int udt_transpose_bf16(const void* in, void* out, int32_t R, int32_t C, int32_t ld, int32_t Rp, void* stream);
/* out[i] (+)= sum_p in[p * n + i], fp32, fixed order */
int udt_reduce_rows_f32(const float* in, float* out, int32_t P, int64_t n, int32_t accumulate, void* stream);
/* row-block partials of the two column reductions below: partials must hold udt_colparts(rows) * C (colsum) / * 2 C (LayerNorm) floats */
int32_t udt_colparts(int64_t rows);
/* out fp32 [C] = column sums of x bf16 [rows, C] (bias gradient of nn.Linear) */
int udt_colsum_bf16(const void* x, float* partials, float* out, int64_t rows, int32_t C, void* stream);
/* LayerNorm parameter gradients: dgamma_dbeta fp32 [2, C] = (sum_r dy * xhat, sum_r dy), x / dy bf16 [rows, C] */
int udt_ln_param_grad(const void* x, const void* dy, float* partials, float* dgamma_dbeta, int64_t rows, int32_t C, float eps, void* stream);
/* weight gradient of nn.Linear (y = x W^T under autograd; attention.py:108-112,193-199 for the trained t_attn projections):
 * dw fp32 [N, K] = dy^T x for dy bf16 [R, N] (row stride ldy), x bf16 [R, K] (row stride ldx); N, K, ldy, ldx multiples of 8, 16-byte
 * aligned operands.  The rows are cut into udt_wgrad_splits(R, N, K) ranges; partials: fp32 scratch of splits * N * K values (may be
 * NULL when splits == 1), summed in range order */
int32_t udt_wgrad_splits(int64_t R, int32_t N, int32_t K);
int udt_wgrad_bf16(const void* dy, const void* x, float* dw, float* partials, int64_t R, int32_t N, int32_t K, int32_t ldy, int32_t ldx,
                   void* stream);
/* text cross-attention, context side (attention.py:140-175 under autograd): dk, dv bf16 [batch * L, lddkv] (head h at columns h * 64)
 * from q bf16 [batch * nq, ldq], v, probs, d_probs (optional), d_o (optional) as udt_xattn_bwd.  The queries are cut into
 * udt_xattn_kv_splits(nq) ranges, one workgroup per (head, sample, range); partials: fp32 scratch of
 * splits * batch * L * heads * 64 * 2 values, summed in ascending range order */
int32_t udt_xattn_kv_splits(int32_t nq);
int udt_xattn_bwd_kv(const void* q, const void* v, const float* probs, const float* d_probs, const void* d_o, void* dk, void* dv,
                     float* partials, int32_t batch, int32_t heads, int32_t head_dim, int32_t nq, int32_t L, int32_t ldq, int32_t ldkv,
                     int32_t ldo, int32_t lddkv, float scale, void* stream);
/* FullLoss.get_local_loss (loss.py:237-286) per layer and its gradient: seg fp32 [B, seg_l, Hs, Ws] character segment maps,
 * seg_mask fp32 [B, seg_l]; d_probs (zero-initialised) += weight * d f / d probs, loss_accum[b] (optional) += f_b */
int udt_local_loss_seg_bwd(const float* probs, const float* seg, const float* seg_mask, const float* gkernel9, float* d_probs,
                           float* loss_accum, float* scratch /* B * seg_l floats: the tokens' terms */, int32_t B, int32_t heads,
                           int32_t size, int32_t L, int32_t seg_l, int32_t Hs, int32_t Ws, float weight, void* stream);
/* eps-prediction loss (loss.py:60-71,131-150; EpsScaling / EpsWeighting): loss fp32 [B] = mean(sigma^-2 (eps * -sigma + noised -
 * target)^2) and d_eps bf16 NHWC [B, hw, cpad] = d mean_b(loss_b) / d eps; eps fp32 NHWC [B, hw, ld_eps], noised / target fp32 NCHW */
int udt_diff_loss_grad(const float* eps, const float* noised, const float* target, const float* sigma, void* d_eps, float* loss, int32_t B,
                       int32_t hw, int32_t ld_eps, int32_t cpad, void* stream);
/* torch.optim.AdamW step on fp32 parameters (diffusion.py:49-51,219): g is scaled by grad_scale first (gradient accumulation / world) */
int udt_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int32_t step, float grad_scale, void* stream);
/* x fp32 += a * y fp32: the attend-and-excite update x <- x - alpha * grad (sampling.py:247) */
int udt_axpy_f32(float* x, const float* y, float a, int64_t n, void* stream);
/* x bf16 += y bf16 (n elements, n % 8 == 0) ; utility for residuals outside GEMM epilogues */
int udt_add_bf16(void* x, const void* y, int64_t n, void* stream);

/* Tuning knobs (no reference counterpart); every setting computes the same results: "gemm_impl" (4 or 8: kernel
 * generation), "conv3p" (0/1: patch-staged 3x3 convolution), "rows_epi" (0/1: row-coalesced epilogues), "n_block"
 * (-1 automatic, 0 off, n: N-tiles per block of the GEMM tile order).  Measurement builds of the library
 * (-DUDT_MEASURE) additionally accept the cost-attribution keys "no_xchg", "no_epi", "no_store", "no_res", "no_bias",
 * "no_fast", which switch parts of the finishing code off and give WRONG results; the product library rejects them. */
int udt_debug_set(const char* key, int32_t value);

/* out[r][c] = bf16(x[r][c] + bias[c]); x/out bf16 [rows, C] (may alias), bias fp32 [C], C % 8 == 0.
 * Used where a projection's GEMM input is identically zero (cross-attention over an all-zero context — the
 * unconditional half under force_uc_zero_embeddings, reference sgm/modules/attention.py:150-174 with k = v = 0 — so
 * to_out(...) + x reduces to bias + x). */
int udt_bias_add_bf16(const void* x, const float* bias, void* out, int64_t rows, int32_t C, void* stream);

/* ---- library services -------------------------------------------------------------------------- */
/* ---- packed-weight handles: the only memory the library allocates on a caller's behalf (SURVEY §8b "Ownership") ----------
 * Checkpoint layouts in (the reference's): nn.Linear weight [N, K] fp32 (+ bias [N]), nn.Conv2d weight [N, Cin, kh, kw] fp32,
 * DEVICE pointers; out: an immutable handle whose buffers have the layouts udt_gemm consumes.
 *   udt_pack_linear: bf16 [Npad][Kpad] (K padded to 64, N to 4, zeros) or, dtype = UDT_DTYPE_FP8_E4M3, e4m3 bytes [Npad][Kpad128]
 *     + colscale[n] = max_k |w[n, k]| / 448; geglu != 0: rows permuted into [32 value | 32 gate] blocks for UDT_GEMM_GEGLU
 *     (GEGLU.proj, reference attention.py:83-99), bias permuted the same way.
 *   udt_pack_conv: bf16 [Npad][kh*kw*Cpad], k = (ky*kw + kx)*Cpad + c; `segments` (may be NULL) = channel counts of the
 *     concatenated sources (decoder skip concat), each padded to a multiple of 64 separately; N padded to n_pad_to.
 * udt_packed_dim(h, which): 0 N, 1 Npad, 2 K, 3 Kpad (= the `ldw` / `K` to put into udt_gemm_desc), 4 dtype.
 * The LayerNorm fold of udt_ln_gemm_fwd (W' = gamma * W, colsum, c) is done by the Python front end (packing.pack_ln_linear). */
typedef struct udt_packed udt_packed;
int udt_pack_linear(const float* w, const float* bias, int32_t N, int32_t K, int32_t dtype, int32_t geglu,
                    udt_packed** out, void* stream);
int udt_pack_conv(const float* w, const float* bias, int32_t N, int32_t Cin, int32_t kh, int32_t kw, const int32_t* segments,
                  int32_t n_segments, int32_t n_pad_to, udt_packed** out, void* stream);
const void* udt_packed_weight(const udt_packed* h);
const float* udt_packed_bias(const udt_packed* h);
const float* udt_packed_colscale(const udt_packed* h);
int32_t udt_packed_dim(const udt_packed* h, int32_t which);
int udt_free_packed(udt_packed* h);

/* ---- the names SURVEY §8b lists for the generic entry points (same functions) -------------------------------------------------
 * udt_gemm_fwd = udt_gemm; udt_conv1x1_fwd = udt_gemm restricted to UDT_GEMM_CONV with ksize 1 (nn.Conv2d 1x1: proj_in / proj_out of
 * the VAE attention block, skip_connection); udt_workspace_bytes = udt_gemm_workspace_bytes; udt_sampler_step = udt_cfg_euler_step
 * (VanillaCFG + DiscreteDenoiser scaling + the Euler update, reference sampling.py:68-80, guiders.py:17-22, denoiser.py:23-31). */
int udt_gemm_fwd(const udt_gemm_desc* d, void* workspace, size_t workspace_bytes, void* stream);
int udt_conv1x1_fwd(const udt_gemm_desc* d, void* workspace, size_t workspace_bytes, void* stream);
size_t udt_workspace_bytes(const udt_gemm_desc* d);
int udt_sampler_step(float* x, const float* eps, float* denoised_out, int32_t B, int32_t hw, int32_t ld_eps,
                     float c_out, float sigma, float sigma_next, float cfg_scale, void* stream);

const char* udt_version(void);
const char* udt_status_string(int status);
int udt_last_hip_error(void);            /* hipError_t of the last failing HIP call (0 if none)     */
int udt_device_arch_ok(void);            /* 1 if device 0 reports gfx950                           */

/* Per-op-class HIP-event timing on the launch stream (used by bench.py's roofline object).
 * classes: 0 conv3x3, 1 gemm (linear/1x1), 2 attn, 3 xattn, 4 norm, 5 elementwise.               */
#define UDT_PROF_NCLASS 6
int udt_prof_enable(uint32_t class_mask);
int udt_prof_reset(void);
/* per-launch trace (class, ms, shape tag) of the profiled classes; dump writes a CSV                  */
int udt_prof_trace(int32_t on);
int udt_prof_dump(const char* path);
/* synchronises the recorded events and returns total ms / launch count for a class               */
int udt_prof_get(int32_t op_class, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* UDT_KERNELS_H */
