"""Parameter containers whose forward is a launch into libudt_kernels.so.

Each container keeps its parameters in the reference's checkpoint layout (fp32, same names/shapes) so that
``load_state_dict`` of a reference ``.ckpt`` / ``.safetensors`` works unchanged, and lazily derives the
device layout the kernels consume (bf16, K-contiguous, channel-padded) in ``packed()``.
Activations between containers are bf16, channel-last: ``[B, H, W, C]`` (or ``[rows, C]`` for tokens).
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
import torch.nn as nn

from udifftext_amd import lib as L
from udifftext_amd import ops, packing

from ..util import init_skipped


def count_flops(kind: str, flops: float) -> None:
    """algorithmic-work accounting for bench.py's roofline objects (ops.WORK_COUNTER)"""
    ops.count_work(kind, flops)


# UDT_FUSE_GN=1: GroupNorm(+SiLU) in front of a 3x3 convolution runs ON THE CONVOLUTION'S STAGED INPUT PATCH whenever the
# producers of the input emitted their column statistics and the library takes the shape (udt_gn_silu_conv3x3_fwd).
# Default OFF: measured on one MI355X box against the separate gn_stats / gn_apply kernels (tools/compare_builds.sh,
# profiles/r02_ab_fused_groupnorm.txt) the fused chain is SLOWER end to end (UNet step 14.3 vs 13.7 ms): the patch
# transform costs ~7 % of a convolution and the per-column statistics in the convolution's accumulator-layout epilogue
# (one pixel per lane: 32-lane reductions) cost more than the two HBM passes they replace.
FUSE_GN = os.environ.get("UDT_FUSE_GN", "0") != "0"

# UDT_GN_EPI (default 1): the GroupNorm STATISTICS come out of the producers' epilogues (the lean GEMM / convolution kernels'
# STATS variants sum the stored values per column in the fp32-row epilogue: ~200 VALU instructions per wave and tile), a tiny
# finalize launch turns them into a scale / shift table and the apply pass stays a separate streaming kernel — the gn_stats
# read of every 64x64 / 32x32 activation is gone, the patch transform that made the fully fused chain slower is not involved.
# The 16x16 / 8x8 levels keep their one-launch strip GroupNorm, which skips its own statistics pass when the producers' are there.
GN_EPI = os.environ.get("UDT_GN_EPI", "1") != "0" and os.environ.get("UDT_LEAN", "") != "0"
EMIT_STATS = FUSE_GN or GN_EPI


def want_stats(B: int, HW: int, C: int) -> bool:
    """should a producer of a [B, HW, C] activation emit column statistics?  (the consumers that read them: the fused chain,
    udt_gn_finalize + udt_gn_apply_scsh, udt_gn_strip_stats)"""
    return FUSE_GN or (GN_EPI and C % 8 == 0)


# UDT_FP8=1 (BASELINE config #5, second generation — round 5): every linear of the transformer blocks whose width is a multiple of
# 128 (the 640- and 1280-channel levels: q|k|v, to_out, GEGLU, ff.net[2], proj_out — and proj_in's OUTPUT) runs on MX8 operands:
# e4m3 weights with per-output-channel scales, e4m3 activations with one E8M0 scale per 32 channels (csrc/common.h "MX8
# activations") that the PRODUCING kernel's epilogue writes next to its bf16 result (lean GEMM, fused text cross-attention, flash
# attention) — no quantisation pass, no normalised copy: the LayerNorms stay folded into the consuming GEMMs, their row statistics
# come out of the producers' epilogues too.  fp32 accumulation, bf16 residual stream.  The 320-channel level (K = 320: five
# K-tiles, epilogue-bound — e4m3 operands buy nothing there, tools/bench_mx8.py), the text cross-attention, convolutions, norm
# statistics and the sampler are unchanged.  (Round 2's first generation — a LayerNorm -> e4m3 kernel with a static per-tensor scale in
# front of the 8-wave fp8 GEMM, LayerNorm-fed linears only — measured SLOWER than the LayerNorm-folded bf16 GEMMs and is gone.)
FP8_LINEARS = os.environ.get("UDT_FP8", "0") != "0"


# ... and attn1's Q K^T and P V (config #5's "fp8 attention"; UDT_FP8_ATTN=0 keeps the bf16 flash kernel for A/B measurements):
# udt_attn_mx8_fwd on v_mfma_scale_f32_32x32x64_f8f6f4.  The q|k|v projection then writes ONLY the MX8 form of its result — q and k
# with block scales along the head dimension, v with one fixed power-of-two multiplier per layer (P V contracts over keys) — at all
# three widths (the 320-channel level through the row-resident kernel's emitting epilogue); the softmax numerators are produced
# directly as e4m3 bytes (csrc/attention.hip), the maxima and the normalisation stay fp32.
FP8_ATTENTION = os.environ.get("UDT_FP8_ATTN", "1") != "0"


def fp8_attention() -> bool:
    return FP8_LINEARS and LN_GEMM and FP8_ATTENTION


def v_fixed_mul(w_v: torch.Tensor, c_v: torch.Tensor) -> float:
    """the fixed e4m3 multiplier of a layer's v projection behind a LayerNorm: v_j = sum_k n_k w_jk + c_j with n a normalised row
    (mean 0, variance 1), so |v_j| is of the order of ||w_j||_2; 448 (the largest e4m3 value; the emitting epilogue saturates) is put
    at 12 ||w_j||_2 + |c_j| of the widest row, rounded down to a power of two (exact to undo).  Data-free, computed once at packing."""
    bound = 12.0 * float(w_v.float().norm(dim=1).max()) + float(c_v.float().abs().max())
    return float(2.0 ** math.floor(math.log2(448.0 / max(bound, 1e-20))))


def mx8_width(c: int) -> bool:
    """does a transformer block of this width run its linears on MX8 operands in config #5?"""
    return FP8_LINEARS and LN_GEMM and c % 128 == 0

# LayerNorm folded into the GEMM that consumes it (udt_ln_gemm_fwd, csrc/lean.h): `attn1(norm1(x))` and `ff(norm3(x))` of
# every transformer block (reference attention.py:310-339) run as ONE launch on the raw residual stream — the normalised
# activation never exists in memory.  UDT_LN_GEMM=0 restores layernorm kernel + GEMM (A/B measurements; config #5's MX8 linears
# need the folded form and are off with it).
LN_GEMM = os.environ.get("UDT_LN_GEMM", "1") != "0" and os.environ.get("UDT_LEAN", "") != "0"


def carry_stats(dst: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    """a reshape / view makes a new tensor object: hand the producer's column statistics over"""
    st = ops.gn_stats_of(src)
    if st is not None:
        dst.gn_stats = st
    return dst


def carry_mx8(dst: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    """the same for the MX8 twin a producer attached to its result (``out.mx8``)"""
    q = ops.mx8_of(src)
    if q is not None:
        dst.mx8 = q
    return dst


def _init_uniform_(p: torch.Tensor, fan_in: int) -> None:
    if init_skipped():
        return
    bound = 1.0 / math.sqrt(max(fan_in, 1))
    with torch.no_grad():
        p.uniform_(-bound, bound)


class _Packed(nn.Module):
    """mixin: cache of repacked device weights, invalidated when parameters move or change.
    ``fused_children()``: child modules whose weights this module's ``_pack`` consumes (fused q|k|v, ...) — they are never
    evaluated on their own, so load-time ``prepare`` does not pack them separately.  After ``prepare(free_masters=True)``
    the cache is frozen (``_pk_frozen``): the fp32 masters are gone and ``packed()`` never looks at them again."""

    def _key(self):
        ps = list(self.parameters(recurse=False))
        return tuple((p.data_ptr(), p._version, str(p.device)) for p in ps)

    def fused_children(self):
        return []

    def packed(self):
        if getattr(self, "_pk_frozen", False):
            if not hasattr(self, "_pk"):
                raise L.UdtError(f"{type(self).__name__}: prepare(free_masters=True) kept only the LayerNorm-folded layout of this "
                                 "module (UDT_LN_GEMM on) and released the fp32 masters — the plain layout cannot be built any "
                                 "more; reload the checkpoint into a fresh engine to change the launch path")
            return self._pk
        key = self._key()
        if getattr(self, "_pk_key", None) != key:
            with torch.no_grad():
                self._pk = self._pack()
            self._pk_key = key
        return self._pk

    def _pack(self):
        raise NotImplementedError

    def packed_fp8(self):
        """the fp8 (e4m3 + per-channel scale) layout of the same weights, cached like ``packed()``"""
        if getattr(self, "_pk8_frozen", False):
            return self._pk8
        if getattr(self, "_pk_frozen", False):
            raise L.UdtError(f"{type(self).__name__}: the fp32 masters were released by prepare(free_masters=True) before the "
                             "fp8 layout was built — call prepare() with UDT_FP8=1 set (it then packs and freezes the e4m3 "
                             "layouts too), or reload the checkpoint")
        key = self._key()
        if getattr(self, "_pk8_key", None) != key:
            with torch.no_grad():
                self._pk8 = self._pack_fp8()
            self._pk8_key = key
        return self._pk8

    def _pack_fp8(self):
        raise NotImplementedError(f"{type(self).__name__} has no fp8 layout")

    def packed_ln(self, norm):
        """the LayerNorm-folded layout (packing.pack_ln_linear) of these weights behind ``norm`` for udt_ln_gemm_fwd:
        (gamma o W as bf16, c = W beta + bias, s = row sums of the folded weights); cached like ``packed()`` and frozen
        with it by ``prepare(free_masters=True)``"""
        if getattr(self, "_pkln_frozen", False):
            return self._pkln
        key = (self._key(), norm.weight.data_ptr(), norm.weight._version, norm.bias.data_ptr(), norm.bias._version)
        if getattr(self, "_pkln_key", None) != key:
            with torch.no_grad():
                self._pkln = self._pack_ln(norm.weight, norm.bias)
            self._pkln_key = key
        return self._pkln

    def _pack_ln(self, gamma, beta):
        raise NotImplementedError(f"{type(self).__name__} has no LayerNorm-folded layout")

    def packed_ln_mx8(self, norm):
        """the LayerNorm-folded layout on e4m3 weights (packing.pack_ln_linear_mx8) for UDT_GEMM_MX8 + ln_colsum: (gamma o W as
        e4m3, per-channel scales, c, s); cached and frozen like ``packed_ln()``"""
        if getattr(self, "_pkln8_frozen", False):
            return self._pkln8
        if getattr(self, "_pk_frozen", False) and getattr(self, "_pkln8_key", None) is None:
            raise L.UdtError(f"{type(self).__name__}: the fp32 masters were released by prepare(free_masters=True) before the MX8 "
                             "layout was built — call prepare() with UDT_FP8=1 set, or reload the checkpoint")
        key = (self._key(), norm.weight.data_ptr(), norm.weight._version, norm.bias.data_ptr(), norm.bias._version)
        if getattr(self, "_pkln8_key", None) != key:
            with torch.no_grad():
                self._pkln8 = self._pack_ln_mx8(norm.weight, norm.bias)
            self._pkln8_key = key
        return self._pkln8

    def _pack_ln_mx8(self, gamma, beta):
        raise NotImplementedError(f"{type(self).__name__} has no LayerNorm-folded MX8 layout")


class Linear(_Packed):
    def __init__(self, in_features: int, out_features: int, bias: bool = True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        _init_uniform_(self.weight, in_features)
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features))
            _init_uniform_(self.bias, in_features)
        else:
            self.register_parameter("bias", None)

    def _pack(self):
        return packing.pack_linear(self.weight), packing.pad_bias(self.bias)

    def _pack_fp8(self):
        wq, cs = packing.pack_linear_fp8(self.weight)
        return wq, cs, packing.pad_bias(self.bias)

    def forward(self, x, residual=None, flags: int = 0, out=None, rowvec=None, rows_per_batch: int = 0,
                colstats: bool = False, x8=None, emit_q8: bool = False, emit_rowstats: bool = False):
        """x8: the input as an MX8 activation (ops.Mx8Act, written by its producer's epilogue) -> the e4m3 GEMM (config #5; x may then
        be None).  emit_q8 (+ emit_rowstats): the result ALSO as an MX8 activation (``out.mx8``) for the next e4m3 GEMM."""
        if x8 is not None:
            wq, cs, b = self.packed_fp8()
            M = x8.data.shape[0]
            if colstats:
                rpb = rows_per_batch if rows_per_batch > 0 else M
                colstats = want_stats(M // rpb, rpb, wq.shape[0])
            return ops.linear_mx8(x8, wq, cs, b, residual=residual, flags=flags, out=out, rows_per_batch=rows_per_batch,
                                  colstats=bool(colstats), emit_q8=emit_q8, emit_rowstats=emit_rowstats)
        w, b = self.packed()
        if colstats:
            rpb = rows_per_batch if rows_per_batch > 0 else x.shape[0]
            colstats = want_stats(x.shape[0] // rpb, rpb, w.shape[0])
        return ops.linear(x, w, b, residual=residual, flags=flags, out=out, rowvec=rowvec, rows_per_batch=rows_per_batch,
                          colstats=bool(colstats), emit_q8=emit_q8, emit_rowstats=emit_rowstats)


class Conv2d(_Packed):
    """3x3 / 1x1 convolution evaluated as implicit GEMM on NHWC bf16 (``segments``: channel counts of the
    concatenated sources this conv reads, for packing)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, padding: int = 0):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        fan_in = in_channels * kernel_size * kernel_size
        _init_uniform_(self.weight, fan_in)
        _init_uniform_(self.bias, fan_in)
        self.segments = None
        self.n_pad = 4          # output channels are padded to a multiple of this (64 when the consumer is a GEMM)
        self.emit_colstats = False      # the output feeds a GroupNorm: emit its statistics from the epilogue

    def _pack(self):
        return packing.pack_conv(self.weight, self.segments, self.n_pad), packing.pad_bias(self.bias, self.n_pad)

    def forward(self, x, x2=None, residual=None, rowvec=None, upsample: bool = False, flags: int = 0,
                pad: Optional[tuple] = None, out_hw: Optional[tuple] = None, norm: Optional["GroupNorm"] = None,
                norm_silu: bool = False, colstats: Optional[bool] = None):
        """norm (+ norm_silu): GroupNorm (+ SiLU) of the (channel-concatenated) input in front of the convolution —
        the reference's ``GroupNorm32 -> SiLU -> conv`` chains (openaimodel.py:183-187,218-231; model.py:128-148).  Fused
        into the convolution (statistics from the producers' epilogues, scale/shift applied on the staged patch) when
        possible, otherwise the separate GroupNorm kernels run first.  colstats: emit the output's statistics."""
        if (x2 is not None and self.segments is None and not getattr(self, "_pk_frozen", False)
                and (x.shape[-1] % 64 != 0 or x2.shape[-1] % 64 != 0)):
            # channel counts of concatenated sources: each is padded to 64 separately, so the layout only differs from the
            # load-time pack (and a repack is only needed) when one of them is not a multiple of 64 — never on this path
            self.segments = [x.shape[-1], x2.shape[-1]]
            self._pk_key = None
        w, b = self.packed()
        if pad is None:
            pad = (self.padding, self.padding)
        if colstats is None:
            colstats = self.emit_colstats
        if colstats:                                    # statistics are emitted where a consumer will use them
            oh, ow = out_hw if out_hw is not None else ((x.shape[1] * (2 if upsample else 1) + 2 * pad[0] - self.kernel_size) // self.stride + 1,
                                                        (x.shape[2] * (2 if upsample else 1) + 2 * pad[1] - self.kernel_size) // self.stride + 1)
            colstats = want_stats(x.shape[0], oh * ow, w.shape[0])
        colstats = bool(colstats)
        kw = dict(ksize=self.kernel_size, stride=self.stride, pad=pad, upsample=upsample, out_hw=out_hw, residual=residual,
                  rowvec=rowvec, flags=flags, n_out=w.shape[0], colstats=colstats)
        in_scsh = None
        if norm is not None:
            st1, st2 = ops.gn_stats_of(x), ops.gn_stats_of(x2)
            if (FUSE_GN and st1 is not None and (x2 is None or st2 is not None)
                    and ops.conv2d(x, w, b, x2=x2, probe_in_scsh=True, **kw)):
                B = x.shape[0]
                in_scsh = ops.gn_finalize(st1, x.shape[-1], st2, x2.shape[-1] if x2 is not None else 0, norm.weight,
                                          norm.bias, B, x.shape[1] * x.shape[2], norm.num_groups, norm.eps)
                count_flops("fused_gn_convs", 1)
            else:
                x, x2 = norm(x, x2=x2, silu=norm_silu), None
        out = ops.conv2d(x, w, b, x2=x2, in_scsh=in_scsh, in_act=1 if norm_silu else 0, **kw)
        if ops.WORK_COUNTER is not None:   # algorithmic (un-padded) multiply-adds x 2
            k = self.kernel_size
            kind = "conv3x3" if k == 3 else "conv1x1"
            npix_out = out.shape[0] * out.shape[1] * out.shape[2]
            count_flops(kind, 2.0 * npix_out * self.out_channels * self.in_channels * k * k)
            # algorithmic HBM bytes: read the input(s) once, the weights once, write the output once (+ residual read)
            nbytes = 2.0 * x.numel() / x.shape[-1] * self.in_channels + 2.0 * self.weight.numel() \
                + npix_out * self.out_channels * (4.0 if (flags & L.GEMM_OUT_F32) else 2.0)
            if residual is not None:
                nbytes += 2.0 * npix_out * self.out_channels
            count_flops(kind + "_bytes", nbytes)
            Hh, Ww = x.shape[1], x.shape[2]
            if (k == 3 and self.stride == 1 and not upsample and tuple(pad) == (1, 1)
                    and self.out_channels > 64 and not (flags & L.GEMM_OUT_F32)
                    and ((Ww % 32 == 0 and Hh % 8 == 0) or (Ww % 16 == 0 and Hh % 16 == 0) or (Ww == 8 and Hh == 8))):
                # the launches udt_gemm routes to c3p::conv3p_kernel (same test as conv3p_geometry in gemm.hip)
                count_flops("conv3p_bytes", nbytes)
                count_flops("conv3p_launches", 1.0)
        return out


class GroupNorm(nn.Module):
    def __init__(self, num_groups: int, num_channels: int, eps: float = 1e-5):
        super().__init__()
        self.num_groups, self.num_channels, self.eps = num_groups, num_channels, eps
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))

    def forward(self, x, x2=None, silu: bool = False):
        if GN_EPI:
            st1, st2 = ops.gn_stats_of(x), ops.gn_stats_of(x2)
            C1, C2 = x.shape[-1], (x2.shape[-1] if x2 is not None else 0)
            HW = x.numel() // (x.shape[0] * C1)
            if (st1 is not None and (x2 is None or st2 is not None)
                    and (ops.gn_strip_ok(x.shape[0], HW, C1, C2, self.num_groups)
                         or ((C1 + C2) % 64 == 0 and (C1 + C2) // self.num_groups <= 128))):
                count_flops("gn_from_epilogue_stats", 1)
                return ops.group_norm_from_stats(x, st1, self.weight, self.bias, self.num_groups, self.eps, silu, x2=x2, st2=st2)
        return ops.group_norm(x, self.weight, self.bias, self.num_groups, self.eps, silu, x2=x2)


class LayerNorm(nn.Module):
    def __init__(self, dim: int, eps: float = 1e-5):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))

    def forward(self, x):
        return ops.layer_norm(x, self.weight, self.bias, self.eps)


def fuse_rows(*weights: torch.Tensor) -> torch.Tensor:
    """stack [Ni, K] matrices row-wise and pack to bf16 (fused q|k, k|v projections)"""
    return packing.pack_linear(torch.cat(list(weights), dim=0))


GEMM_GEGLU = L.GEMM_GEGLU
GEMM_OUT_F32 = L.GEMM_OUT_F32
GEMM_RELU = L.GEMM_RELU
GEMM_TRANSPOSED = L.GEMM_TRANSPOSED
GEMM_SILU_OUT = L.GEMM_SILU_OUT
