"""Transformer blocks of the UNet on the gfx950 kernels (tokens are bf16 ``[B, N, C]``).

State-dict names and semantics follow reference sgm/modules/attention.py:
  GEGLU / FeedForward              :44-70    -> one GEMM with a fused x*gelu(gate) epilogue + one GEMM
  CrossAttention (t_attn)          :111-174  -> q GEMM, hoistable k|v GEMM on the context, short-context kernel
                                               that can emit the softmax probabilities (attn_map_cache)
  MemoryEfficientCrossAttention    :177-262  -> fused q|k GEMM, V^T GEMM (transposed epilogue), flash attention
  BasicTransformerBlock            :265-341  -> pre-LN residual blocks; residual adds fused in GEMM epilogues
  SpatialTransformer               :344-415  -> GroupNorm(eps 1e-6), proj_in/proj_out linears; NHWC activations
                                               make the reference's two permute copies (:405,:412) disappear
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.nn as nn

from udifftext_amd import ops, packing

from . import hipnn as H
from .diffusionmodules.util import zero_module


# The text cross-attention branch runs as ONE udt_tattn_fused launch on per-batch folded tables (csrc/tattn.hip); the
# reference's op sequence (layernorm -> to_q GEMM -> short-context attention -> to_out GEMM) remains for the calls that must
# return attention probabilities (noise search) and for contexts the tables do not cover (1 token, > 12 tokens).
TATTN_FUSED = True          # (tests switch it off to compare the two forms)

# UDT_FF_PROJ (default 1, round 6): the last block's ff.net[2] and the SpatialTransformer's proj_out run as ONE GEMM.  Both are linear
# and nothing sits between them but a residual add (reference attention.py:69-70,336-339,409-411):
#     proj_out(ff.net[2](h) + t3) + x  =  [h | t3] [W_po W_2 | W_po]^T + (W_po b_2 + b_po) + x
# — the product W_po W_2 is formed once, in fp32, when the weights are packed; the launch is the two-source 1x1 form of the lean GEMM
# (the ResBlocks' skip_connection kernel) with K = 4 C + C, the same FLOPs as the two launches, one launch / one [M, C] round trip
# less per transformer (16 per UNet call), and no bf16 rounding of the intermediate.  Config #5's MX8 chain keeps its two launches.
FF_PROJ = os.environ.get("UDT_FF_PROJ", "1") != "0"


class GEGLU(H._Packed):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = H.Linear(dim_in, dim_out * 2)

    def _key(self):
        return self.proj._key()

    def fused_children(self):
        return [self.proj]

    def _pack(self):
        return packing.pack_geglu(self.proj.weight, self.proj.bias)

    def _pack_ln(self, gamma, beta):
        return packing.pack_ln_linear(self.proj.weight, self.proj.bias, gamma, beta, geglu=True)

    def _pack_ln_mx8(self, gamma, beta):
        return packing.pack_ln_linear_mx8(self.proj.weight, self.proj.bias, gamma, beta, geglu=True)

    def forward(self, x, ln=None, x8=None):
        """ln: the LayerNorm in front of this projection, folded into the GEMM (x is then the RAW activation).
        x8 (with ln): the raw activation as an MX8 activation with row statistics -> the e4m3 GEMM; the hidden activation is then
        returned as an ops.Mx8Act ONLY (config #5: its one consumer is the e4m3 ff.net[2])"""
        if ln is not None and x8 is not None:
            wq, cs, c, s = self.packed_ln_mx8(ln)
            return ops.linear_mx8(x8, wq, cs, ln_c=c, ln_s=s, eps=ln.eps, flags=H.GEMM_GEGLU, emit_q8=True, want_bf16=False)
        if ln is not None:
            wf, c, s = self.packed_ln(ln)
            return ops.ln_linear(x, wf, c, s, eps=ln.eps, flags=H.GEMM_GEGLU)
        w, b = self.packed()
        return ops.linear(x, w, b, flags=H.GEMM_GEGLU)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, glu=True, dropout=0.0):
        super().__init__()
        if not glu:
            raise NotImplementedError("only the gated (GEGLU) feed-forward is on the UDiffText path")
        inner = int(dim * mult)
        self.net = nn.Sequential(GEGLU(dim, inner), nn.Identity(), H.Linear(inner, dim_out or dim))

    def forward(self, x, residual=None, ln=None, x8=None, emit_q8: bool = False, emit_rowstats: bool = False):
        h = self.net[0](x, ln=ln, x8=x8)
        if isinstance(h, ops.Mx8Act):                       # (config #5: e4m3 hidden activation -> e4m3 ff.net[2])
            return self.net[2](None, residual=residual, x8=h, emit_q8=emit_q8, emit_rowstats=emit_rowstats)
        return self.net[2](h, residual=residual)


class CrossAttention(H._Packed):
    """text cross-attention over L <= 16 context tokens"""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner = dim_head * heads
        context_dim = context_dim or query_dim
        self.scale = dim_head ** -0.5
        self.heads, self.dim_head = heads, dim_head
        self.to_q = H.Linear(query_dim, inner, bias=False)
        self.to_k = H.Linear(context_dim, inner, bias=False)
        self.to_v = H.Linear(context_dim, inner, bias=False)
        self.to_out = zero_module(nn.Sequential(H.Linear(inner, query_dim), nn.Identity()))
        self.attn_map_cache = None

    def _key(self):
        return self.to_k._key() + self.to_v._key()

    def fused_children(self):
        return [self.to_k, self.to_v]

    def _pack(self):
        return H.fuse_rows(self.to_k.weight, self.to_v.weight)

    def project_context(self, context_bf16: torch.Tensor) -> torch.Tensor:
        """context [B, L, Dc] bf16 -> k|v [B, L, 2*inner]; step-invariant, hoisted by the sampler"""
        B, Lc, Dc = context_bf16.shape
        return ops.linear(context_bf16.reshape(B * Lc, Dc), self.packed()).reshape(B, Lc, -1)

    def forward(self, x, context=None, kv=None, residual=None, emit_map: bool = False, out=None):
        """x: bf16 [B, N, C]"""
        inner = self.heads * self.dim_head
        B, N, _ = x.shape
        q = self.to_q(x.reshape(B * N, -1)).reshape(B, N, inner)
        if kv is None:
            kv = self.project_context(context)
        probs = None
        if emit_map and self.attn_map_cache is not None:
            probs = torch.empty((B * self.heads, N, kv.shape[1]), dtype=torch.float32, device=q.device)
            self.attn_map_cache["size"] = int(N ** 0.5)
            self.attn_map_cache["attn_map"] = probs
        o = ops.xattention(q, kv[..., :inner], kv[..., inner:], self.heads, self.dim_head, self.scale, probs=probs)
        res = residual.reshape(B * N, -1) if residual is not None else None
        out2 = out.reshape(B * N, -1) if out is not None else None
        return self.to_out[0](o.reshape(B * N, inner), residual=res, out=out2).reshape(B, N, -1)

    def prepare_fused(self, kv: torch.Tensor, t_norm, out=None):
        """fold the (step-invariant) context k|v, to_q, to_out and the LayerNorm in front of this module into the
        per-sample tables of udt_tattn_fused (None for a single-token context: the reference then applies a sigmoid
        instead of the softmax, attention.py:159-162 — that case stays on the unfused path)"""
        if kv.shape[1] < 2 or kv.shape[1] > 12 or self.dim_head != 64 or not TATTN_FUSED:
            return None              # (udt_tattn_prepare takes 2..12 context tokens of head_dim 64: anything else -> xattn chain)
        wq, _ = self.to_q.packed()
        wo, _ = self.to_out[0].packed()
        return ops.tattn_prepare(kv.contiguous(), wq, wo, t_norm.weight, t_norm.bias, self.heads, self.scale, out=out)

    def zero_context_residual(self, x, out):
        """x + t_attn(anything, context == 0): with k = v = 0 (to_k / to_v have no bias) the attention output is 0
        and to_out reduces to its bias — bit-identical to running the projections on zeros."""
        _, b = self.to_out[0].packed()
        return ops.bias_add(x, b, out=out)


# One q|k|v GEMM per self-attention (reference attention.py:193-199 applies three bias-free Linears to the same input) and
# the flash kernel reads V row-major (udt_attn_rowv_fwd).  (Round 2's q|k GEMM + transposed-output V GEMM form is gone.)


class MemoryEfficientCrossAttention(H._Packed):
    """self-attention (flash kernel, head_dim 64)"""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0, **kwargs):
        super().__init__()
        if dim_head != 64:
            raise NotImplementedError("udt_attn_fwd is specialised for head_dim 64 (num_head_channels: 64)")
        inner = dim_head * heads
        context_dim = context_dim or query_dim
        self.heads, self.dim_head = heads, dim_head
        self.to_q = H.Linear(query_dim, inner, bias=False)
        self.to_k = H.Linear(context_dim, inner, bias=False)
        self.to_v = H.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(H.Linear(inner, query_dim), nn.Identity())

    def _key(self):
        return self.to_q._key() + self.to_k._key() + self.to_v._key()

    def fused_children(self):
        return [self.to_q, self.to_k, self.to_v]

    def _pack(self):
        # one q|k|v projection; the flash kernel transposes V tiles out of LDS itself
        return H.fuse_rows(self.to_q.weight, self.to_k.weight, self.to_v.weight), None

    def _pack_ln(self, gamma, beta):
        pk = packing.pack_ln_linear(torch.cat([self.to_q.weight, self.to_k.weight, self.to_v.weight], 0), None, gamma, beta)
        inner = self.heads * self.dim_head
        self.v_mul = H.v_fixed_mul(pk[0][2 * inner:3 * inner], pk[1][2 * inner:3 * inner])     # (for the e4m3 attention)
        return pk

    def _pack_ln_mx8(self, gamma, beta):
        pk = packing.pack_ln_linear_mx8(torch.cat([self.to_q.weight, self.to_k.weight, self.to_v.weight], 0), None, gamma, beta)
        inner = self.heads * self.dim_head
        wv = pk[0][2 * inner:3 * inner].view(torch.float8_e4m3fn).float() * pk[1][2 * inner:3 * inner, None]
        self.v_mul8 = H.v_fixed_mul(wv, pk[2][2 * inner:3 * inner])
        return pk

    def forward(self, x, context=None, mask=None, residual=None, ln=None, x8=None):
        """ln: the LayerNorm in front of the q|k|v projection, folded into that GEMM (x is then the RAW activation).
        x8 (with ln; config #5): x as an MX8 activation with row statistics -> e4m3 q|k|v GEMM; the flash kernel then writes its
        output as MX8 too and to_out runs on e4m3 operands.  With hipnn.FP8_ATTENTION the projection writes q|k|v as MX8 ONLY and
        the attention itself runs on e4m3 operands (ops.attention_mx8)."""
        if context is not None or mask is not None:
            raise NotImplementedError("attn1 is pure self-attention on this path (reference attention.py:251-252)")
        inner = self.heads * self.dim_head
        B, N, C = x.shape
        x2 = x.reshape(B * N, C)
        mx = ln is not None and x8 is not None
        a8 = ln is not None and H.fp8_attention() and N % 4 == 0
        scale = self.dim_head ** -0.5
        o = None
        if mx:
            wq, cs, c, sv = self.packed_ln_mx8(ln)
            if a8:
                qkv8 = ops.linear_mx8(x8, wq, cs, ln_c=c, ln_s=sv, eps=ln.eps, emit_q8=True, want_bf16=False,
                                      q8_fixed=(2 * inner, self.v_mul8))
                o = ops.attention_mx8(qkv8, B, self.heads, scale, self.v_mul8, emit_q8=True)
            else:
                qkv = ops.linear_mx8(x8, wq, cs, ln_c=c, ln_s=sv, eps=ln.eps).reshape(B, N, 3 * inner)
        elif ln is not None:
            wf, c, sv = self.packed_ln(ln)
            qkv8 = None
            if a8 and C == 320:                  # (the row-resident K = 320 kernel has the emitting epilogue)
                # (None: that kernel is switched off — udt_debug_set("rowres", 0), UDT_LEAN=1 / 6 — or x is not 16-byte aligned: the
                #  level then keeps the bf16 projection + bf16 flash attention, like sampling with UDT_FP8_ATTN=0 does)
                qkv8 = ops.ln_linear(x2, wf, c, sv, eps=ln.eps, emit_q8=True, want_bf16=False, q8_fixed=(2 * inner, self.v_mul),
                                     q8_or_none=True)
            if qkv8 is not None:
                o = ops.attention_mx8(qkv8, B, self.heads, scale, self.v_mul)
            else:
                qkv = ops.ln_linear(x2, wf, c, sv, eps=ln.eps).reshape(B, N, 3 * inner)
        else:
            qkv = ops.linear(x2, self.packed()[0]).reshape(B, N, 3 * inner)
        if o is None:
            o = ops.attention_rowv(qkv[..., :inner], qkv[..., inner:2 * inner], qkv[..., 2 * inner:], self.heads, scale, emit_q8=mx)
        res = residual.reshape(B * N, -1) if residual is not None else None
        return self.to_out[0](o.reshape(B * N, inner), residual=res, x8=ops.mx8_of(o) if mx else None).reshape(B, N, -1)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, d_head, dropout=0.0, t_context_dim=None, v_context_dim=None, gated_ff=True):
        super().__init__()
        self.attn1 = MemoryEfficientCrossAttention(query_dim=dim, heads=n_heads, dim_head=d_head)
        if t_context_dim is not None and t_context_dim > 0:
            self.t_attn = CrossAttention(query_dim=dim, context_dim=t_context_dim, heads=n_heads, dim_head=d_head)
            self.t_norm = H.LayerNorm(dim)
        if v_context_dim is not None and v_context_dim > 0:
            raise NotImplementedError("v_attn is not configured in UDiffText (v_context_dim unset)")
        self.norm1 = H.LayerNorm(dim)
        self.norm3 = H.LayerNorm(dim)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)

    def fused_tattn_ok(self, n_tokens: int) -> bool:
        C = self.t_attn.heads * self.t_attn.dim_head
        return (TATTN_FUSED and self.t_attn.dim_head == 64 and C in (320, 640, 1280) and self.t_attn.heads * 64 == C
                and n_tokens % (32 if C >= 640 else 64) == 0)       # (the kernel's token tile: csrc/tattn.hip tattn_tt)

    def forward(self, x, t_context=None, v_context=None, t_kv=None, emit_map: bool = False, zero_ctx_rows: int = 0,
                t_fused=None, x8=None, emit_rowstats: bool = False, defer_ff_out: bool = False):
        """zero_ctx_rows: the first n samples of the batch attend to an all-zero text context (the unconditional
        half of a CFG pair under force_uc_zero_embeddings) — their t_attn branch is x + to_out.bias, no GEMMs.
        t_fused: the block's folded context tables (ops.TattnTables for the samples of x) -> one fused launch.
        x8 (config #5): x as an MX8 activation with row statistics (written by the producer of x) -> the block's linears run on
        e4m3 operands and the result carries its own MX8 twin (``.mx8``; with row statistics if emit_rowstats: the next consumer
        is another block's LayerNorm-folded q|k|v)"""
        fold = H.LN_GEMM                                        # LayerNorm inside the consuming GEMM (udt_ln_gemm_fwd)
        mx = fold and x8 is not None and H.mx8_width(x.shape[-1])
        x = self.attn1(x, residual=x, ln=self.norm1, x8=x8 if mx else None) if fold else self.attn1(self.norm1(x), residual=x)
        if hasattr(self, "t_attn"):
            n0 = 0 if emit_map else min(int(zero_ctx_rows), x.shape[0])
            if t_fused is not None and not emit_map and n0 < x.shape[0] and self.fused_tattn_ok(x.shape[1]):
                _, bo = self.t_attn.to_out[0].packed()
                x = ops.tattn_fused(x, t_fused, bo, self.t_attn.heads, n0, self.t_norm.eps, emit_q8=mx)
            elif n0 > 0 and t_kv is not None:
                y = torch.empty_like(x)
                self.t_attn.zero_context_residual(x[:n0], y[:n0])
                if n0 < x.shape[0]:
                    xc = x[n0:]
                    self.t_attn(self.t_norm(xc), kv=t_kv[n0:], residual=xc, out=y[n0:])
                x = y
            else:
                x = self.t_attn(self.t_norm(x), context=t_context, kv=t_kv, residual=x, emit_map=emit_map)
        B, N, C = x.shape
        x2 = x.reshape(B * N, C)
        if fold:
            x8b = ops.mx8_of(x) if mx else None               # (None: the unfused text cross-attention ran — bf16 feed-forward)
            if defer_ff_out and x8b is None:
                # FF_PROJ: hand (GEGLU hidden activation, the rows the feed-forward's residual adds) to the SpatialTransformer, whose
                # ONE launch applies ff.net[2], the residual and proj_out
                return self.ff.net[0](x2, ln=self.norm3), x2
            out = self.ff(x2, residual=x2, ln=self.norm3, x8=x8b, emit_q8=x8b is not None, emit_rowstats=emit_rowstats)
            return H.carry_mx8(out.reshape(B, N, C), out)
        return self.ff(self.norm3(x2), residual=x2).reshape(B, N, C)

    def prepare_ln(self, freeze: bool = False) -> int:
        """build (and optionally freeze) the LayerNorm-folded layouts — and, in config #5, the e4m3 layouts of the block's linears
        (the bf16 ones stay: calls that must return attention maps take the unfused text cross-attention and a bf16 feed-forward);
        returns their bytes"""
        n = 0
        for mod, norm in ((self.attn1, self.norm1), (self.ff.net[0], self.norm3)):
            for t in mod.packed_ln(norm):
                n += t.numel() * t.element_size()
            if freeze:
                mod._pkln_frozen = True
        if H.mx8_width(self.norm1.dim):
            for mod, norm in ((self.attn1, self.norm1), (self.ff.net[0], self.norm3)):
                for t in mod.packed_ln_mx8(norm):
                    n += t.numel() * t.element_size()
                if freeze:
                    mod._pkln8_frozen = True
            for lin in (self.attn1.to_out[0], self.ff.net[2]):
                for t in lin.packed_fp8():
                    if t is not None:
                        n += t.numel() * t.element_size()
                if freeze:
                    lin._pk8_frozen = True
        return n


class SpatialTransformer(nn.Module):
    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, t_context_dim=None, v_context_dim=None,
                 use_linear=False):
        super().__init__()
        if not use_linear:
            raise NotImplementedError("use_linear_in_transformer: True is the UDiffText configuration")
        self.in_channels = in_channels
        inner = n_heads * d_head
        self.norm = H.GroupNorm(32, in_channels, eps=1e-6)
        self.proj_in = H.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, t_context_dim=t_context_dim,
                                   v_context_dim=v_context_dim) for _ in range(depth)])
        self.proj_out = zero_module(H.Linear(inner, in_channels))
        self.use_linear = use_linear

    def project_context(self, context_bf16) -> List[torch.Tensor]:
        return [blk.t_attn.project_context(context_bf16) for blk in self.transformer_blocks]

    def prepare_fused(self, kv_list, out=None) -> list:
        """per block: fold its hoisted k|v (+ to_q, to_out, t_norm) into the tables of udt_tattn_fused"""
        return [blk.t_attn.prepare_fused(kv, blk.t_norm, out=(out[i] if out is not None else None))
                for i, (blk, kv) in enumerate(zip(self.transformer_blocks, kv_list))]

    def forward(self, x, t_context=None, v_context=None, t_kv: Optional[list] = None, emit_map: bool = False,
                zero_ctx_rows: int = 0, t_fused: Optional[list] = None):
        """x: bf16 NHWC [B, H, W, C]"""
        B, Hh, Ww, C = x.shape
        N = Hh * Ww
        # config #5: proj_in's epilogue also writes its result as an MX8 activation + the row statistics of norm1 (hipnn.FP8_LINEARS)
        mx = H.mx8_width(self.proj_in.out_features)
        t2 = self.proj_in(self.norm(x).reshape(B * N, C), emit_q8=mx, emit_rowstats=mx)
        t = H.carry_mx8(t2.reshape(B, N, -1), t2)
        nb = len(self.transformer_blocks)
        defer = FF_PROJ and H.LN_GEMM and not mx
        for i, blk in enumerate(self.transformer_blocks):
            t = blk(t, t_context=t_context, t_kv=(t_kv[i] if t_kv is not None else None), emit_map=emit_map,
                    zero_ctx_rows=zero_ctx_rows, t_fused=(t_fused[i] if t_fused is not None else None),
                    x8=ops.mx8_of(t) if mx else None, emit_rowstats=(i + 1 < nb), defer_ff_out=(defer and i + 1 == nb))
        if isinstance(t, tuple):
            hid, t3 = t
            w, b = self.packed_ffproj()
            st = H.want_stats(B, N, w.shape[0])
            out = ops.conv2d(hid.reshape(B, Hh, Ww, -1), w, b, x2=t3.reshape(B, Hh, Ww, -1), ksize=1, pad=(0, 0), residual=x,
                             n_out=w.shape[0], colstats=st)
            if ops.WORK_COUNTER is not None:          # (algorithmic work of the two linears it replaces)
                inner = hid.shape[-1]
                H.count_flops("gemm", 2.0 * B * N * C * (inner + C))
                H.count_flops("gemm_bytes", 2.0 * B * N * (inner + 3 * C) + 2.0 * C * (inner + C))
                H.count_flops("gemm_launches", 1.0)
            return out
        # the output feeds a ResBlock's GroupNorm (and maybe a skip concat): statistics from the GEMM epilogue
        out = self.proj_out(t.reshape(B * N, -1), residual=x.reshape(B * N, C), rows_per_batch=N, colstats=True,
                            x8=ops.mx8_of(t) if mx else None)
        return H.carry_stats(out.reshape(B, Hh, Ww, C), out)

    def packed_ffproj(self):
        """[W_po W_2 | W_po] as a two-source 1x1-convolution weight (bf16 [C, 4 C + C]) and the bias W_po b_2 + b_po; cached by
        the two modules' parameter versions, frozen by prepare(free_masters=True)"""
        if getattr(self, "_pkfp_frozen", False):
            return self._pkfp
        lin2, po = self.transformer_blocks[-1].ff.net[2], self.proj_out
        key = (lin2._key(), po._key())
        if getattr(self, "_pkfp_key", None) != key:
            with torch.no_grad():
                w2, wpo = lin2.weight.float(), po.weight.float()
                wf = torch.cat([wpo @ w2, wpo], dim=1)
                bias = wpo @ lin2.bias.float() + po.bias.float()
                self._pkfp = (packing.pack_conv(wf[:, :, None, None], [w2.shape[1], wpo.shape[1]]), packing.pad_bias(bias))
            self._pkfp_key = key
        return self._pkfp

    def prepare_ffproj(self, freeze: bool = False) -> int:
        n = 0
        if FF_PROJ and H.LN_GEMM:
            for t in self.packed_ffproj():
                n += t.numel() * t.element_size()
            if freeze:
                self._pkfp_frozen = True
        return n

    def prepare_mx8(self, freeze: bool = False) -> int:
        """config #5: the e4m3 layout of proj_out (proj_in stays a bf16 GEMM: its input is a GroupNorm output); bytes"""
        n = 0
        if H.mx8_width(self.proj_in.out_features):
            for t in self.proj_out.packed_fp8():
                if t is not None:
                    n += t.numel() * t.element_size()
            if freeze:
                self.proj_out._pk8_frozen = True
        return n
