"""Weight repacking from the reference's checkpoint layouts (fp32, NCHW conv kernels, [out,in] linears)
into the device layouts the gfx950 kernels consume (bf16, K-contiguous rows, K padded to 64).

Checkpoint key names/shapes are the reference's (SURVEY.md §8b "Checkpoint contract"); packing happens
after ``load_state_dict`` in each module's ``prepare()``.
"""
from __future__ import annotations

import torch

KPAD = 64


def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def pad_channels(c: int) -> int:
    return _round_up(c, KPAD)


def pack_linear(w: torch.Tensor, n_pad_to: int = 4) -> torch.Tensor:
    """[N, K] fp32 -> bf16 [Npad, Kpad] (zero padded)."""
    N, K = w.shape
    Np, Kp = _round_up(N, n_pad_to), _round_up(K, KPAD)
    out = torch.zeros((Np, Kp), dtype=torch.bfloat16, device=w.device)
    out[:N, :K] = w.to(torch.bfloat16)
    return out.contiguous()


def pack_conv(w: torch.Tensor, segments=None, n_pad_to: int = 4) -> torch.Tensor:
    """[N, Cin, kh, kw] fp32 -> bf16 [Npad, kh*kw*Cpad], k = (ky*kw + kx)*Cpad + c.

    ``segments`` lists the channel counts of concatenated sources; each is padded to 64 separately
    (the UNet's skip concats are all multiples of 64 already, so this is the identity there)."""
    N, Cin, kh, kw = w.shape
    if segments is None:
        segments = [Cin]
    assert sum(segments) == Cin
    Np = _round_up(N, n_pad_to)
    parts = []
    c0 = 0
    for seg in segments:
        sp = pad_channels(seg)
        blk = torch.zeros((Np, kh, kw, sp), dtype=torch.bfloat16, device=w.device)
        blk[:N, :, :, :seg] = w[:, c0:c0 + seg].permute(0, 2, 3, 1).to(torch.bfloat16)
        parts.append(blk)
        c0 += seg
    out = torch.cat(parts, dim=3) if len(parts) > 1 else parts[0]
    return out.reshape(Np, -1).contiguous()


def pad_bias(b: torch.Tensor | None, n_pad_to: int = 4) -> torch.Tensor | None:
    if b is None:
        return None
    N = b.shape[0]
    Np = _round_up(N, n_pad_to)
    out = torch.zeros((Np,), dtype=torch.float32, device=b.device)
    out[:N] = b.float()
    return out


def geglu_permutation(inner: int) -> torch.Tensor:
    """Row permutation for a GEGLU projection [2*inner, C]: packed rows come in blocks of 64 =
    [32 value rows | the 32 matching gate rows] so that one wave's accumulator holds both
    (gemm.hip, UDT_GEMM_GEGLU).  ``inner`` must be a multiple of 32."""
    assert inner % 32 == 0
    blocks = inner // 32
    idx = torch.empty((blocks, 2, 32), dtype=torch.long)
    base = torch.arange(32)
    for k in range(blocks):
        idx[k, 0] = 32 * k + base
        idx[k, 1] = inner + 32 * k + base
    return idx.reshape(-1)


def pack_geglu(w: torch.Tensor, b: torch.Tensor):
    """GEGLU.proj weight [2*inner, C] / bias [2*inner] -> permuted bf16 weight, fp32 bias."""
    inner = w.shape[0] // 2
    perm = geglu_permutation(inner).to(w.device)
    return pack_linear(w[perm]), b[perm].float().contiguous()


def pack_ln_linear(w: torch.Tensor, b, gamma: torch.Tensor, beta: torch.Tensor, geglu: bool = False):
    """LayerNorm(gamma, beta) followed by a linear [N, K] (+ bias), folded for udt_ln_gemm_fwd:
        LN(x) W^T + b = rstd * (x W'^T - mean * s) + c,   W' = W * gamma (bf16),  s_n = sum_k W'_nk (of the ROUNDED
    weights, so the mean term cancels exactly what the MFMA accumulates),  c_n = sum_k beta_k W_nk + b_n.
    Returns (W' bf16 [N, Kpad], c fp32 [N], s fp32 [N]); GEGLU projections are row-permuted like pack_geglu."""
    wf = w.float()
    bf = torch.zeros((w.shape[0],), dtype=torch.float32, device=w.device) if b is None else b.float()
    if geglu:
        perm = geglu_permutation(w.shape[0] // 2).to(w.device)
        wf, bf = wf[perm], bf[perm]
    wp = pack_linear(wf * gamma.float()[None, :])
    s = wp[:, :w.shape[1]].float().sum(dim=1).contiguous()
    c = pad_bias((wf @ beta.float() + bf).contiguous())          # (padded like W' and s: the kernel reads 4-wide c and s vectors)
    if c.shape[0] != wp.shape[0]:
        raise ValueError(f"pack_ln_linear: {w.shape[0]} output rows are not a multiple of the bias padding")
    return wp, c, s


# ------------------------------------------------------------------------------------------------ fp8 (config #5)
FP8_KPAD = 128          # K elements per 128-byte LDS row of the fp8 GEMM
FP8_MAX = 448.0         # largest finite OCP e4m3 value


def pack_linear_fp8(w: torch.Tensor, n_pad_to: int = 4):
    """[N, K] fp32 -> (e4m3 bytes [Npad, Kpad128] as uint8, per-output-channel scale fp32 [Npad]):
    w[n, k] ~= q[n, k] * scale[n] with scale[n] = max_k |w[n, k]| / 448 (per-channel weight scales, udt_gemm colscale) —
    the weight side of an UDT_GEMM_MX8 launch (the activation side carries E8M0 block scales written by its producer)."""
    N, K = w.shape
    Np, Kp = _round_up(N, n_pad_to), _round_up(K, FP8_KPAD)
    wf = w.float()
    amax = wf.abs().amax(dim=1).clamp_min(1e-12)
    scale = amax / FP8_MAX
    q = (wf / scale[:, None]).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).view(torch.uint8)
    out = torch.zeros((Np, Kp), dtype=torch.uint8, device=w.device)
    out[:N, :K] = q
    cs = torch.ones((Np,), dtype=torch.float32, device=w.device)
    cs[:N] = scale
    return out.contiguous(), cs


def pack_ln_linear_mx8(w: torch.Tensor, b, gamma: torch.Tensor, beta: torch.Tensor, geglu: bool = False):
    """pack_ln_linear for the MX8 path (udt_gemm_desc UDT_GEMM_MX8 + ln_colsum): W' = W * gamma as e4m3 with per-output-channel
    scales, s_n = sum_k of the DEQUANTISED W' (so that the mean term cancels what the fp8 MFMA accumulates), c_n = sum_k beta_k W_nk
    + b_n.  Returns (W'q uint8 [N, Kpad128], colscale fp32 [N], c fp32 [N], s fp32 [N]); GEGLU rows permuted like pack_geglu."""
    wf = w.float()
    bf = torch.zeros((w.shape[0],), dtype=torch.float32, device=w.device) if b is None else b.float()
    if geglu:
        perm = geglu_permutation(w.shape[0] // 2).to(w.device)
        wf, bf = wf[perm], bf[perm]
    wq, cs = pack_linear_fp8(wf * gamma.float()[None, :])
    s = (wq.view(torch.float8_e4m3fn).float() * cs[:, None]).sum(dim=1).contiguous()
    c = pad_bias((wf @ beta.float() + bf).contiguous())
    if c.shape[0] != wq.shape[0]:
        raise ValueError(f"pack_ln_linear_mx8: {w.shape[0]} output rows are not a multiple of the bias padding")
    return wq, cs, c, s
