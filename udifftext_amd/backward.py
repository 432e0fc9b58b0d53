"""Attend-and-excite on the HIP path: d local_loss / d x through the UNet (reference sampling.py:233-252).

The reference calls ``torch.autograd.grad(local_loss, x)`` on the graph autograd recorded through ``model.model(x, c_noise, cond)``.
Here the network's kernels are launches into libudt_kernels.so, so the reverse pass is written out: every layer type of the
UNet has a ``*_fwd`` that runs the forward through the same ops as inference (unfused where the fused inference kernels keep no
intermediate: LayerNorm + GEMM instead of the folded form, GEMM + udt_geglu_fwd instead of the GEGLU epilogue, the xattn chain
that emits probabilities instead of the fused text attention), keeps what the reverse pass needs, and returns a closure
``bwd(d_out) -> d_in``.  Only activations are differentiated (x is what attend-and-excite updates; the weights are frozen):

  * linear / convolution backward-data = the FORWARD GEMM / convolution kernels on re-packed weights (``W^T``; the 3x3 taps
    rotated by 180 degrees with the channel roles swapped), cached per module;
  * a stride-2 convolution's backward-data = the stride-1 convolution of the zero-dilated gradient with those weights;
  * nearest x2 upsampling backward = 2 x 2 block sums (udt_sum2x2_bf16);
  * flash attention, text cross-attention, local loss, GroupNorm (+ SiLU), LayerNorm, GEGLU: csrc/backward.hip.

A cotangent of ``None`` means "exactly zero": layers downstream of the last t_attn map the loss reads are never differentiated.
torch is used for memory only (allocation, views, channel concatenation / split, the zero-dilation copy).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops, packing

Bwd = Callable[[Optional[torch.Tensor]], Optional[torch.Tensor]]
DEBUG_SUMS: Optional[list] = None       # (tools/debug_train3.py: per-layer checksums of the tape-mode forward)

_CACHE: dict = {}


def _cached(mod, tag: str, build):
    key = (id(mod), tag)
    ver = mod._key()
    hit = _CACHE.get(key)
    if hit is None or hit[0] != ver:
        with torch.no_grad():
            hit = (ver, build())
        _CACHE[key] = hit
    return hit[1]


def clear_cache() -> None:
    _CACHE.clear()


def _need_masters(mod) -> None:
    if getattr(mod, "_pk_frozen", False):
        raise ops.L.UdtError("attend-and-excite needs the fp32 master weights for the backward layouts: build the engine without "
                             "prepare(free_masters=True)")


# ------------------------------------------------------------------------------------------------ linear / convolution
def linear_bwd(lin, dy: torch.Tensor, add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dX = dY W (+ add): the forward GEMM on W^T packed as [in_features, out_features]"""
    _need_masters(lin)
    wt = _cached(lin, "wT", lambda: packing.pack_linear(lin.weight.detach().t().contiguous()))
    return ops.linear(dy, wt, None, residual=add)


def _conv_wt(conv, n_pad: int = 4) -> torch.Tensor:
    """backward-data weights of a convolution: Wb[ci, co, ky, kx] = W[co, ci, k - 1 - ky, k - 1 - kx], packed like a forward conv
    that reads ``out_channels`` channels and writes ``in_channels``"""
    _need_masters(conv)
    return _cached(conv, f"wT{n_pad}", lambda: packing.pack_conv(conv.weight.detach().flip(2, 3).permute(1, 0, 2, 3).contiguous(), None, n_pad))


def conv_bwd(conv, dy: torch.Tensor, add: Optional[torch.Tensor] = None, n_pad: int = 4) -> torch.Tensor:
    """backward-data of a stride-1 'same' convolution (3x3 pad 1, or 1x1)"""
    assert conv.stride == 1 and conv.kernel_size in (1, 3)
    wt = _conv_wt(conv, n_pad)
    return ops.conv2d(dy, wt, None, ksize=conv.kernel_size, stride=1, pad=(conv.kernel_size // 2,) * 2, residual=add, n_out=wt.shape[0])


def down_bwd(conv, dy: torch.Tensor, in_hw: Tuple[int, int]) -> torch.Tensor:
    """backward-data of the stride-2 3x3 pad-1 convolution (Downsample.op): zero-dilate dY to the input size, stride-1 conv"""
    assert conv.stride == 2 and conv.kernel_size == 3 and conv.padding == 1
    B, Ho, Wo, Cc = dy.shape
    z = torch.zeros((B, in_hw[0], in_hw[1], Cc), dtype=dy.dtype, device=dy.device)
    z[:, 0:2 * Ho:2, 0:2 * Wo:2] = dy                    # (memory only: z[2 o] = dY[o])
    wt = _conv_wt(conv)
    return ops.conv2d(z, wt, None, ksize=3, stride=1, pad=(1, 1), n_out=wt.shape[0])


# ------------------------------------------------------------------------------------------------ layers
def resblock_fwd(rb, x: torch.Tensor, emb_rows: torch.Tensor, x2: Optional[torch.Tensor] = None):
    """ResBlock (reference openaimodel.py:163-250) on bf16 NHWC; x2: the skip source concatenated on channels (decoder).
    Returns (out, bwd) with bwd(d_out) -> (d_x, d_x2)"""
    C1 = x.shape[-1]
    xc = x if x2 is None else torch.cat([x, x2], dim=-1)
    gn1, conv1 = rb.in_layers[0], rb.in_layers[2]
    gn2, conv2 = rb.out_layers[0], rb.out_layers[3]
    emb_out = emb_rows[:, rb.emb_offset:rb.emb_offset + rb.out_channels]
    a1 = ops.group_norm(xc, gn1.weight, gn1.bias, gn1.num_groups, gn1.eps, True)
    w1, b1 = conv1.packed()
    h1 = ops.conv2d(a1, w1, b1, ksize=3, rowvec=emb_out, n_out=w1.shape[0])
    a2 = ops.group_norm(h1, gn2.weight, gn2.bias, gn2.num_groups, gn2.eps, True)
    ident = isinstance(rb.skip_connection, nn.Identity)
    if ident:
        skip = xc
    else:
        ws, bs = rb.skip_connection.packed()
        skip = ops.conv2d(xc, ws, bs, ksize=1, pad=(0, 0), n_out=ws.shape[0])
    w2, b2 = conv2.packed()
    out = ops.conv2d(a2, w2, b2, ksize=3, residual=skip, n_out=w2.shape[0])
    del a1, a2, skip

    def bwd(d_out):
        if d_out is None:
            return None, None
        d_a2 = conv_bwd(conv2, d_out)
        d_h1 = ops.group_norm_bwd(h1, d_a2, gn2.weight, gn2.bias, gn2.num_groups, gn2.eps, True)
        d_a1 = conv_bwd(conv1, d_h1)
        d_skip = d_out if ident else conv_bwd(rb.skip_connection, d_out)
        d_xc = ops.group_norm_bwd(xc, d_a1, gn1.weight, gn1.bias, gn1.num_groups, gn1.eps, True, add=d_skip)
        if x2 is None:
            return d_xc, None
        return d_xc[..., :C1].contiguous(), d_xc[..., C1:].contiguous()
    return out, bwd


def transformer_block_fwd(blk, t1: torch.Tensor, B: int, kv: torch.Tensor, rec: list, name: str, kv_c: Optional[torch.Tensor] = None):
    """BasicTransformerBlock (reference attention.py:286-339) on bf16 rows t1 [B * N, C]; kv: the hoisted context projection
    [B, L, 2 C]; kv_c: the projection of the CENTRED context (ops.center_tokens) — its k half replaces kv's in the forward (the
    softmax over the tokens does not see a common shift of the keys) and both halves serve the reverse pass (nor does dS see a common
    shift of the values), see csrc/backward.hip center_tokens_kernel; the t_attn probabilities go to ``rec`` (a dict per map; its
    ``d_probs`` is filled in before the reverse pass)"""
    M, Cc = t1.shape
    N = M // B
    a1, ta, ff = blk.attn1, blk.t_attn, blk.ff
    heads = a1.heads
    scale = a1.dim_head ** -0.5
    n1 = ops.layer_norm(t1, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
    qkv = ops.linear(n1, a1.packed()[0]).reshape(B, N, 3 * Cc)
    o = ops.attention_rowv(qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:], heads, scale)
    t2 = a1.to_out[0](o.reshape(M, Cc), residual=t1)
    n2 = ops.layer_norm(t2, blk.t_norm.weight, blk.t_norm.bias, blk.t_norm.eps)
    q = ta.to_q(n2).reshape(B, N, Cc)
    probs = torch.empty((B * ta.heads, N, kv.shape[1]), dtype=torch.float32, device=t1.device)
    if kv_c is None or kv.shape[1] < 2:                              # (one token: the sigmoid branch is not shift-invariant)
        kv_c = kv
    kk, vv, vv_c = kv_c[..., :Cc], kv[..., Cc:], kv_c[..., Cc:]
    o2 = ops.xattention(q, kk, vv, ta.heads, ta.dim_head, ta.scale, probs=probs)
    # ``pgrads``: set to a dict before the reverse pass to ALSO collect the gradients of this block's trainable parameters (the
    # reference trains t_attn / t_norm only, configs/train/textdesign_sd_2.yaml:4-6), keyed by their state-dict names below ``name``;
    # ``ctx`` / ``ctx_c``: the context rows [B * L, Dc] (plain / centred) behind kv, needed for d to_k / d to_v
    item = {"name": name, "heads": ta.heads, "size": int(N ** 0.5), "attn_map": probs, "d_probs": None, "pgrads": None,
            "ctx": None, "ctx_c": None, "block": blk}
    rec.append(item)
    t3 = ta.to_out[0](o2.reshape(M, Cc), residual=t2)
    n3 = ops.layer_norm(t3, blk.norm3.weight, blk.norm3.bias, blk.norm3.eps)
    proj = ff.net[0].proj
    ag = proj(n3)                                                   # stored pre-activations [M, 2 * inner]
    t4 = ff.net[2](ops.geglu(ag), residual=t3)
    del n1, n2, n3

    def param_grads(pg: dict, d_t3, d_o2, dq, d_n2, dP):
        """gradients of t_attn.{to_q, to_k, to_v, to_out.0} and t_norm (fp32, state-dict names)"""
        pre = name + "."                                            # "...transformer_blocks.i.t_attn."
        nrm = name[:-len("t_attn")] + "t_norm."
        if d_t3 is not None:                                        # t3 = o2 W_out^T + b + t2
            pg[pre + "to_out.0.weight"] = ops.weight_grad(d_t3, o2.reshape(M, Cc))
            pg[pre + "to_out.0.bias"] = ops.colsum(d_t3)
        else:
            pg[pre + "to_out.0.weight"] = torch.zeros((Cc, Cc), dtype=torch.float32, device=t1.device)
            pg[pre + "to_out.0.bias"] = torch.zeros((Cc,), dtype=torch.float32, device=t1.device)
        n2r = ops.layer_norm(t2, blk.t_norm.weight, blk.t_norm.bias, blk.t_norm.eps)
        pg[pre + "to_q.weight"] = ops.weight_grad(dq.reshape(M, Cc), n2r)
        dk, dv = ops.xattention_bwd_kv(q, vv_c, probs, dP, d_o2, ta.heads, ta.scale)
        Lc = kv.shape[1]
        cx = item["ctx"]
        cxc = item["ctx_c"] if item["ctx_c"] is not None else cx
        # (sum_l dK_l = 0: the centred rows give the same product, without the cancellation of the tokens' common part)
        pg[pre + "to_k.weight"] = ops.weight_grad(dk.reshape(B * Lc, Cc), cxc)
        pg[pre + "to_v.weight"] = ops.weight_grad(dv.reshape(B * Lc, Cc), cx)
        dg, db = ops.layer_norm_param_grad(t2, d_n2, blk.t_norm.eps)
        pg[nrm + "weight"], pg[nrm + "bias"] = dg, db

    def bwd(d_t4):
        dP = item["d_probs"]
        if d_t4 is None and dP is None:
            return None
        d_t3 = None
        if d_t4 is not None:
            d_hgl = linear_bwd(ff.net[2], d_t4)
            d_ag = ops.geglu_bwd(ag, d_hgl)
            d_n3 = linear_bwd(proj, d_ag)
            d_t3 = ops.layer_norm_bwd(t3, d_n3, blk.norm3.weight, blk.norm3.eps, add=d_t4)
        d_o2 = linear_bwd(ta.to_out[0], d_t3).reshape(B, N, Cc) if d_t3 is not None else None
        dq = ops.xattention_bwd(kk, vv_c, probs, dP, d_o2, ta.heads, ta.scale)
        d_n2 = linear_bwd(ta.to_q, dq.reshape(M, Cc))
        if item["pgrads"] is not None:
            param_grads(item["pgrads"], d_t3, d_o2, dq, d_n2, dP)
        d_t2 = ops.layer_norm_bwd(t2, d_n2, blk.t_norm.weight, blk.t_norm.eps, add=d_t3)
        d_o = linear_bwd(a1.to_out[0], d_t2).reshape(B, N, Cc)
        d_qkv = ops.attention_bwd(qkv, o, d_o, heads, scale)
        _need_masters(a1.to_q)
        wqkv_t = _cached(a1, "wqkvT", lambda: packing.pack_linear(
            torch.cat([a1.to_q.weight, a1.to_k.weight, a1.to_v.weight], 0).detach().t().contiguous()))
        d_n1 = ops.linear(d_qkv.reshape(M, 3 * Cc), wqkv_t, None)
        return ops.layer_norm_bwd(t1, d_n1, blk.norm1.weight, blk.norm1.eps, add=d_t2)
    return t4, bwd


def spatial_transformer_fwd(st, x: torch.Tensor, kv_list: list, rec: list, name: str, kv_c_list: Optional[list] = None):
    """SpatialTransformer (reference attention.py:342-411, use_linear) on bf16 NHWC"""
    B, Hh, Ww, Cc = x.shape
    M = B * Hh * Ww
    g0 = ops.group_norm(x, st.norm.weight, st.norm.bias, st.norm.num_groups, st.norm.eps, False)
    t = st.proj_in(g0.reshape(M, Cc))
    del g0
    bwds = []
    for i, blk in enumerate(st.transformer_blocks):
        t, b = transformer_block_fwd(blk, t, B, kv_list[i], rec, f"{name}transformer_blocks.{i}.t_attn",
                                     kv_c=(kv_c_list[i] if kv_c_list is not None else None))
        bwds.append(b)
    out = st.proj_out(t, residual=x.reshape(M, Cc)).reshape(B, Hh, Ww, Cc)

    def bwd(d_out):
        d_t = linear_bwd(st.proj_out, d_out.reshape(M, Cc)) if d_out is not None else None
        for b in reversed(bwds):
            d_t = b(d_t)
        if d_t is None:
            return d_out
        d_g0 = linear_bwd(st.proj_in, d_t).reshape(B, Hh, Ww, Cc)
        return ops.group_norm_bwd(x, d_g0, st.norm.weight, st.norm.bias, st.norm.num_groups, st.norm.eps, False, add=d_out)
    return out, bwd


def _block_fwd(unet, block, prefix: str, h: torch.Tensor, emb_rows: torch.Tensor, x2, t_kv, rec: list, t_kv_c=None):
    """one TimestepEmbedSequential; returns (out, bwd) with bwd(d_out) -> (d_in, d_x2)"""
    from sgm.modules.attention import SpatialTransformer
    from sgm.modules.diffusionmodules.openaimodel import Downsample, ResBlock, Upsample
    from sgm.modules import hipnn as H
    bwds = []
    for j, layer in enumerate(block):
        if isinstance(layer, ResBlock):
            h, b = resblock_fwd(layer, h, emb_rows, x2=x2)
            x2 = None
            bwds.append(("res", b))
        elif isinstance(layer, SpatialTransformer):
            h, b = spatial_transformer_fwd(layer, h, t_kv[layer.st_index], rec, f"{prefix}{j}.",
                                           kv_c_list=(t_kv_c[layer.st_index] if t_kv_c is not None else None))
            bwds.append(("st", b))
        elif isinstance(layer, Downsample):
            hw = (h.shape[1], h.shape[2])
            w, bb = layer.op.packed()
            h = ops.conv2d(h, w, bb, ksize=3, stride=2, pad=(1, 1), n_out=w.shape[0])
            bwds.append(("lin", lambda d, c=layer.op, hw=hw: down_bwd(c, d, hw) if d is not None else None))
        elif isinstance(layer, Upsample):
            w, bb = layer.conv.packed()
            h = ops.conv2d(h, w, bb, ksize=3, upsample=True, n_out=w.shape[0])
            bwds.append(("lin", lambda d, c=layer.conv: ops.sum2x2(conv_bwd(c, d)) if d is not None else None))
        elif isinstance(layer, H.Conv2d):                             # input_blocks.0: conv3x3 on the 64-channel padded input
            w, bb = layer.packed()
            h = ops.conv2d(h, w, bb, ksize=3, n_out=w.shape[0])
            bwds.append(("lin", lambda d, c=layer: conv_bwd(c, d, n_pad=64) if d is not None else None))
        else:
            raise NotImplementedError(type(layer).__name__)
        if DEBUG_SUMS is not None:
            DEBUG_SUMS.append((f"{prefix}{j}", float(h.float().abs().sum())))

    def bwd(d):
        d_x2 = None
        for kind, b in reversed(bwds):
            if kind == "res":
                d, dx2 = b(d)
                if dx2 is not None:
                    d_x2 = dx2
            else:
                d = b(d)
        return d, d_x2
    return h, bwd


def _acc(a: Optional[torch.Tensor], b: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if a is None:
        return b
    if b is None:
        return a
    return ops.add_(a.contiguous(), b.contiguous())


class UNetTape:
    """Tape-mode forward of the UNet (every block through the ``*_fwd`` functions above) that can be reversed:
        tape = UNetTape(unet, xin, timesteps, t_context, with_head=...)      # xin: bf16 NHWC [B, h, w, CPAD]
        tape.maps         the t_attn probability maps in module order (dicts: name, heads, size, attn_map, d_probs, pgrads)
        tape.eps          fp32 NHWC [B, h, w, 4] when with_head (out.0 GroupNorm + SiLU, out.2 convolution), else None
        tape.backward(d_eps=None, param_grads=None) -> d xin (bf16 NHWC)     # d_eps: bf16 NHWC [B, h, w, 64] cotangent of eps;
                          map cotangents are whatever the caller stored in maps[i]["d_probs"]; param_grads: a dict that receives the
                          fp32 gradients of the t_attn / t_norm parameters under their state-dict names (prefix ``param_prefix``)"""

    def __init__(self, unet, xin: torch.Tensor, timesteps: torch.Tensor, t_context: torch.Tensor, with_head: bool = False,
                 param_prefix: str = "model.diffusion_model."):
        self.unet, self.prefix = unet, param_prefix
        emb_rows = unet.time_embedding_rows(timesteps)
        ctx32 = t_context.float().contiguous()
        t_kv = unet.project_context(ctx32)
        ctx_c = ops.center_tokens(ctx32) if ctx32.shape[1] > 1 else None
        t_kv_c = unet.project_context(ctx_c) if ctx_c is not None else None
        rec: list = []
        self.tape: List = []
        hs = []
        h = xin
        for i, block in enumerate(unet.input_blocks):
            h, b = _block_fwd(unet, block, f"input_blocks.{i}.", h, emb_rows, None, t_kv, rec, t_kv_c)
            self.tape.append(b)
            hs.append(h)
        self.n_in = len(hs)
        h, self.b_mid = _block_fwd(unet, unet.middle_block, "middle_block.", h, emb_rows, None, t_kv, rec, t_kv_c)
        self.out_tape = []
        for i, block in enumerate(unet.output_blocks):
            h, b = _block_fwd(unet, block, f"output_blocks.{i}.", h, emb_rows, hs.pop(), t_kv, rec, t_kv_c)
            self.out_tape.append(b)
        self.maps = rec
        Bc, Lc, Dc = ctx32.shape
        cx = ctx32.to(torch.bfloat16).reshape(Bc * Lc, Dc)
        cxc = ctx_c.reshape(Bc * Lc, Dc) if ctx_c is not None else None
        for it in rec:
            it["ctx"], it["ctx_c"] = cx, cxc
        self.eps = None
        self._head = None
        if with_head:
            gn, conv = unet.out[0], unet.out[2]
            a = ops.group_norm(h, gn.weight, gn.bias, gn.num_groups, gn.eps, True)
            w, bb = conv.packed()
            self.eps = ops.conv2d(a, w, bb, ksize=3, flags=ops.L.GEMM_OUT_F32, n_out=w.shape[0])
            self._head = (h, gn, conv)

    def backward(self, d_eps: Optional[torch.Tensor] = None, param_grads: Optional[dict] = None, debug: Optional[dict] = None):
        for it in self.maps:
            it["pgrads"] = param_grads
        d = None
        if d_eps is not None:
            h, gn, conv = self._head
            d_a = conv_bwd(conv, d_eps)
            d = ops.group_norm_bwd(h, d_a, gn.weight, gn.bias, gn.num_groups, gn.eps, True)
        n_in = self.n_in
        d_skips: List[Optional[torch.Tensor]] = [None] * n_in
        if debug is not None:                                         # (tools/debug_aae.py: maps and block-boundary cotangents)
            debug["maps"] = self.maps
        for j in reversed(range(len(self.out_tape))):
            if debug is not None and d is not None:
                debug[f"d_output_blocks.{j}"] = d.clone()
            d, d_x2 = self.out_tape[j](d)
            d_skips[n_in - 1 - j] = d_x2                              # output block j consumed hs[n_in - 1 - j]
        if debug is not None and d is not None:
            debug["d_middle_block"] = d.clone()
        d, _ = self.b_mid(d)
        for i in reversed(range(n_in)):
            d = _acc(d, d_skips[i])
            if debug is not None and d is not None:
                debug[f"d_input_blocks.{i}"] = d.clone()
            d, _ = self.tape[i](d)
        if param_grads is not None:
            # blocks the reverse pass never reached (downstream of the last map read, no eps cotangent) have zero gradients
            for it in self.maps:
                blk = it["block"]
                pre, nrm = it["name"] + ".", it["name"][:-len("t_attn")] + "t_norm."
                for key, par in ((pre + "to_q.weight", blk.t_attn.to_q.weight), (pre + "to_k.weight", blk.t_attn.to_k.weight),
                                 (pre + "to_v.weight", blk.t_attn.to_v.weight), (pre + "to_out.0.weight", blk.t_attn.to_out[0].weight),
                                 (pre + "to_out.0.bias", blk.t_attn.to_out[0].bias), (nrm + "weight", blk.t_norm.weight),
                                 (nrm + "bias", blk.t_norm.bias)):
                    if key not in param_grads:
                        param_grads[key] = torch.zeros(par.shape, dtype=torch.float32, device=par.device)
            for k in [k for k in param_grads if not k.startswith(self.prefix)]:
                param_grads[self.prefix + k] = param_grads.pop(k)
        return d


def unet_maps_vjp(unet, x: torch.Tensor, timesteps: torch.Tensor, concat: torch.Tensor, t_context: torch.Tensor, maps_grad,
                  debug: Optional[dict] = None) -> torch.Tensor:
    """d F / d x (fp32 [B, 4, h, w]) for a scalar F of the UNet's t_attn probability maps: the tape-mode forward of
    UNet(cat(x, concat)) records every map as a dict (name, heads, size, attn_map fp32 [B * heads, n, L], d_probs None);
    ``maps_grad(maps)`` sets ``d_probs`` = d F / d attn_map on the maps F reads (None elsewhere: those layers, and everything
    downstream of the last one read, are not differentiated); then the reverse pass runs."""
    from sgm.modules.diffusionmodules.openaimodel import CPAD
    xin = ops.nchw_to_nhwc(torch.cat((x.float(), concat.float()), dim=1).contiguous(), CPAD)
    tape = UNetTape(unet, xin, timesteps, t_context, with_head=False)
    maps_grad(tape.maps)
    d = tape.backward(debug=debug)
    if d is None:
        raise ValueError("maps_grad set no d_probs: nothing to differentiate")
    return ops.nhwc_to_nchw(d.contiguous(), 4)


def unet_local_loss_grad(unet, loss_fn, x: torch.Tensor, timesteps: torch.Tensor, concat: torch.Tensor, t_context: torch.Tensor,
                         mask: torch.Tensor, seg_mask: torch.Tensor, debug: Optional[dict] = None):
    """(local_loss fp32 [B], d sum(local_loss) / d x fp32 [B, 4, h, w]) for x fp32 NCHW: the UNet sees cat(x, concat) and the
    t_attn maps of size >= loss_fn.min_attn_size are scored by get_min_local_loss (reference sampling.py:233-252, loss.py:192-235)"""
    B = x.shape[0]
    maskf, seg = mask.float().contiguous(), seg_mask.float().contiguous()
    gk = loss_fn.g_kernel[0, 0].reshape(9).float().contiguous()
    loss = torch.zeros((B,), dtype=torch.float32, device=x.device)
    count = [0]

    def maps_grad(rec):
        used = [it for it in rec if it["size"] >= loss_fn.min_attn_size]
        if not used:
            raise ValueError("no t_attn map reaches loss_fn.min_attn_size: the local loss is undefined for this latent size")
        for it in used:
            it["d_probs"] = torch.zeros_like(it["attn_map"])
            ops.local_loss_bwd(it["attn_map"], maskf, seg, gk, it["d_probs"], loss, it["heads"], it["size"], 1.0 / len(used))
        count[0] = len(used)
    grad = unet_maps_vjp(unet, x, timesteps, concat, t_context, maps_grad, debug=debug)
    return loss / count[0], grad


# ------------------------------------------------------------------------------------------------ hipGraph replay
class GraphedLocalLossGrad:
    """hipGraph replay of ``unet_local_loss_grad`` for ONE set of shapes.  An attend-and-excite gradient is ≈ 2 100 launches whose
    arguments depend on the shapes only (the timestep index, the latent and the conditioning are device data), and the sampler asks for
    one to twenty-one of them before EVERY denoising step (reference sampling.py:240-252, 379-386): the whole evaluation — tape-mode
    forward, loss seeds, reverse pass — is captured once into a hipGraph on a private memory pool (the tape's activations live there)
    and replayed; the host issues one graph launch instead of the kernel launches (B = 1: the eager evaluation is host-bound).
    The inputs are copied into static buffers; ``loss`` / ``grad`` are static outputs, valid until the next call."""

    def __init__(self, unet, loss_fn, x, timesteps, concat, t_context, mask, seg_mask):
        from sgm.modules.diffusionmodules.sampling import weights_fingerprint
        self.unet, self.loss_fn = unet, loss_fn
        self.key = self.key_of(x, timesteps, concat, t_context, mask, seg_mask)
        self.fingerprint = weights_fingerprint(unet)
        mk = lambda t: t.detach().float().contiguous().clone()
        self.inputs = [mk(t) for t in (x, timesteps, concat, t_context, mask, seg_mask)]
        dev = x.device
        self.ws = ops.Workspace(dev)
        self.pool = torch.cuda.graph_pool_handle()
        self.stream = torch.cuda.Stream(device=dev)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.loss = self.grad = None

    @staticmethod
    def key_of(*tensors) -> tuple:
        return tuple((tuple(t.shape), str(t.device)) for t in tensors)

    def valid_for(self, unet, *tensors) -> bool:
        from sgm.modules.diffusionmodules.sampling import weights_fingerprint
        return unet is self.unet and self.key == self.key_of(*tensors) and self.fingerprint == weights_fingerprint(unet)

    def _evaluate(self):
        with ops.launch_context(cu_share=1, workspace=self.ws):
            return unet_local_loss_grad(self.unet, self.loss_fn, *self.inputs)

    def _capture(self) -> None:
        torch.cuda.synchronize()
        with torch.cuda.stream(self.stream):                              # eager pass: weight re-packs, kernel attributes, library pages
            self._evaluate()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self.pool, stream=self.stream, capture_error_mode="thread_local"):
            self.loss, self.grad = self._evaluate()
        self.graph = g

    def __call__(self, x, timesteps, concat, t_context, mask, seg_mask):
        for dst, src in zip(self.inputs, (x, timesteps, concat, t_context, mask, seg_mask)):
            dst.copy_(src)
        if self.graph is None:
            self._capture()
        self.graph.replay()
        return self.loss, self.grad

    def check(self) -> None:
        self.ws.check()
