"""Per-rounding error budget of one UNet call (CPU, the oracle): which bf16 rounding of the HIP path owns how much of the ~1.2e-2
rel RMS the GPU tests measure against the fp32 reference?  Not a test — a script (``python tests/error_budget.py``) that re-runs
oracle.nets.unet_forward on the G7 golden's inputs with bf16 roundings injected at the places where the kernels round:

  W     weights of every conv / linear -> bf16                                   (packing.pack_linear / pack_conv)
  A     the activation operand of every conv / linear -> bf16                    (MFMA operands are bf16)
  O     the OUTPUT of every conv / linear (after bias, + time embedding, + residual where the epilogue adds it) -> bf16
        = the residual stream and every intermediate tensor is stored as bf16
  N     the output of every GroupNorm(+SiLU) / LayerNorm -> bf16                 (gn_apply writes bf16; folded LayerNorms never exist
        in memory — N_ln off reproduces that)
  P     attention: q, k, v as bf16 (O of their GEMM), probabilities -> bf16 before P V, output -> bf16
  L8    BASELINE config #5's linears: in every transformer block whose width is a multiple of 128 (640 / 1280 channels) q|k|v, to_out,
        GEGLU, ff.net[2] and proj_out run on MX8 operands — the activation as e4m3 with one power-of-two scale per 32 channels of a
        row (tests/mx8_ref.py = the kernels' rule), the weight as e4m3 with one scale per output channel (packing.pack_linear_fp8);
        the LayerNorm-fed ones in the folded form on the RAW rows (packing.pack_ln_linear_mx8), statistics from the unquantised rows
  R32   like O, but the residual stream stays fp32: the sums x + h of ResBlock / attention / feed-forward / SpatialTransformer are
        not rounded (what an fp32 residual stream would buy)

Each variant's eps is compared with the fp32 run (rel RMS).  Uses the synthetic weights and the G7 inputs (32x32 latents, CFG pair).
Test infrastructure (imports oracle/): never part of the product path.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nets, sampling, spec          # noqa: E402
from udifftext_amd import synth                  # noqa: E402

bf = lambda t: t.to(torch.bfloat16).float()
FLAGS = dict(W=False, A=False, O=False, N=False, N_ln=True, P=False, R32=False, A8=False, L8=False)


def _e4m3(t, scale):
    """fixed-scale e4m3 round trip"""
    return (t * scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() / scale


def _mx8_lastdim(t):
    """MX8 round trip with 32-element blocks along the last dimension (tests/mx8_ref.py: the kernels' scale rule)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import mx8_ref
    shp = t.shape
    flat = t.reshape(-1, shp[-1])
    pad = (-flat.shape[1]) % 128
    if pad:
        flat = F.pad(flat, (0, pad))
    q, sc = mx8_ref.encode(flat)
    return mx8_ref.decode(q, sc)[:, :shp[-1]].reshape(shp)


def _q(flag, t):
    return bf(t) if FLAGS[flag] else t


# ---- oracle building blocks with the roundings injected (same arithmetic as oracle/nets.py otherwise) -------------------------
def _gn(sd, p, x, eps):
    return F.group_norm(x.float(), 32, sd[p + "weight"], sd[p + "bias"], eps)


def _conv(sd, p, x, stride=1, padding=1, add=None):
    y = F.conv2d(_q("A", x), _q("W", sd[p + "weight"]), sd[p + "bias"], stride=stride, padding=padding)
    if add is not None:
        y = y + add
    return _q("O", y)


def _lin(sd, p, x, bias=True, add=None, keep32=False):
    y = F.linear(_q("A", x), _q("W", sd[p + "weight"]), sd[p + "bias"] if bias else None)
    if add is not None:
        y = y + add
    return y if keep32 else _q("O", y)


def _w8(w):
    """per-output-channel e4m3 round trip of a weight matrix (packing.pack_linear_fp8)"""
    scale = w.abs().amax(dim=1, keepdim=True).clamp_min(1e-12) / 448.0
    return (w / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * scale


def _lin8(sd, p, x, bias=True, add=None, keep32=False, ln=None):
    """a linear of config #5 (flag L8): MX8 activation x e4m3 weight, fp32 accumulation.  ln = (gamma, beta, eps): the LayerNorm in
    front of it is folded in — the GEMM multiplies the RAW rows: LN(x) W^T + b = rstd (x8 W'^T - mean s) + c"""
    w = sd[p + "weight"]
    b = sd[p + "bias"] if bias else None
    shp = x.shape
    x2 = x.reshape(-1, shp[-1])
    if ln is not None:
        gamma, beta, eps = ln
        mean = x2.mean(dim=1, keepdim=True)
        rstd = (x2.var(dim=1, unbiased=False, keepdim=True) + eps).rsqrt()
        w8 = _w8(w * gamma[None, :])
        c = w @ beta + (b if b is not None else 0.0)
        y = rstd * (_mx8_lastdim(x2) @ w8.t() - mean * w8.sum(dim=1)[None, :]) + c[None, :]
    else:
        y = _mx8_lastdim(x2) @ _w8(w).t()
        if b is not None:
            y = y + b
    y = y.reshape(shp[:-1] + (w.shape[0],))
    if add is not None:
        y = y + add
    return y if keep32 else _q("O", y)


def _resblock(sd, p, x, emb):
    e = F.linear(F.silu(emb), sd[p + "emb_layers.1.weight"], sd[p + "emb_layers.1.bias"])[:, :, None, None]   # fp32 rows on the GPU too
    h = _conv(sd, p + "in_layers.2.", _q("N", F.silu(_gn(sd, p + "in_layers.0.", x, 1e-5))), add=e)
    skip = _conv(sd, p + "skip_connection.", x, padding=0) if (p + "skip_connection.weight") in sd else x
    n2 = _q("N", F.silu(_gn(sd, p + "out_layers.0.", h, 1e-5)))
    if FLAGS["R32"]:                                              # fp32 residual stream: the sum is not rounded
        y = F.conv2d(_q("A", n2), _q("W", sd[p + "out_layers.3.weight"]), sd[p + "out_layers.3.bias"], padding=1)
        return skip + y
    return _conv(sd, p + "out_layers.3.", n2, add=skip)


def _self_attention(sd, p, x, heads, res, raw=None, ln=None):
    """raw / ln (flag L8 at an MX8 width): the un-normalised rows and the LayerNorm folded into the e4m3 q|k|v projection"""
    l8 = raw is not None
    lin_out = (lambda pp, t, **kw: _lin8(sd, pp, t, **kw)) if l8 else (lambda pp, t, **kw: _lin(sd, pp, t, **kw))
    if l8:
        q, k, v = (nets._split_heads(_lin8(sd, p + n, raw, bias=False, ln=ln), heads) for n in ("to_q.", "to_k.", "to_v."))
    else:
        q, k, v = (nets._split_heads(_lin(sd, p + n, x, bias=False), heads) for n in ("to_q.", "to_k.", "to_v."))
    d = q.shape[-1]
    if FLAGS["A8"]:
        # an e4m3 attention kernel as sketched in DESIGN.md section 10: q and k with MX blocks along d (two per head), v with one
        # power-of-two scale, the un-normalised probabilities (<= 2^8 under the deferred rescale) as e4m3
        q, k = _mx8_lastdim(q), _mx8_lastdim(k)
        v = _e4m3(v, 2.0 ** float(torch.floor(torch.log2(448.0 / v.abs().max()))))
        sc = q @ k.transpose(-1, -2) * d ** -0.5
        pr = torch.exp(sc - sc.amax(dim=-1, keepdim=True))
        pr8 = _e4m3(pr, 256.0)
        attn_v = (pr8 @ v) / pr.sum(dim=-1, keepdim=True)
        o = _q("P", nets._merge_heads(attn_v))
        return lin_out(p + "to_out.0.", o, add=res, keep32=FLAGS["R32"])
    attn = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, dim=-1)
    o = _q("P", nets._merge_heads(_q("P", attn) @ v))
    return lin_out(p + "to_out.0.", o, add=res, keep32=FLAGS["R32"])


def _text_cross_attention(sd, p, x, ctx, heads, res):
    q = nets._split_heads(_lin(sd, p + "to_q.", x, bias=False), heads)
    k = nets._split_heads(_lin(sd, p + "to_k.", ctx, bias=False), heads)
    v = nets._split_heads(_lin(sd, p + "to_v.", ctx, bias=False), heads)
    sim = (q @ k.transpose(-1, -2) * q.shape[-1] ** -0.5).softmax(dim=-1)
    return _lin(sd, p + "to_out.0.", nets._merge_heads(sim @ v), add=res, keep32=FLAGS["R32"])


def _transformer_block(sd, p, x, ctx, heads):
    c = x.shape[-1]
    ln = lambda n, t: _q("N" if not FLAGS["N_ln"] else "N_off", F.layer_norm(t, (c,), sd[p + n + ".weight"], sd[p + n + ".bias"], 1e-5))
    if FLAGS["L8"] and c % 128 == 0:
        fold = lambda n: (sd[p + n + ".weight"], sd[p + n + ".bias"], 1e-5)
        x = _self_attention(sd, p + "attn1.", None, heads, x, raw=x, ln=fold("norm1"))
        x = _text_cross_attention(sd, p + "t_attn.", ln("t_norm", x), ctx, heads, x)               # (bf16 in config #5)
        val, gate = _lin8(sd, p + "ff.net.0.proj.", x, keep32=True, ln=fold("norm3")).chunk(2, dim=-1)
        return _lin8(sd, p + "ff.net.2.", _q("O", val * F.gelu(gate)), add=x, keep32=FLAGS["R32"])
    x = _self_attention(sd, p + "attn1.", ln("norm1", x), heads, x)
    x = _text_cross_attention(sd, p + "t_attn.", ln("t_norm", x), ctx, heads, x)
    val, gate = _lin(sd, p + "ff.net.0.proj.", ln("norm3", x), keep32=True).chunk(2, dim=-1)      # GEGLU in the epilogue: one rounding
    return _lin(sd, p + "ff.net.2.", _q("O", val * F.gelu(gate)), add=x, keep32=FLAGS["R32"])


def _spatial_transformer(sd, p, x, ctx, heads):
    b, c, h, w = x.shape
    t = _q("N", _gn(sd, p + "norm.", x, 1e-6)).permute(0, 2, 3, 1).reshape(b, h * w, c)
    t = _lin(sd, p + "proj_in.", t)
    t = _transformer_block(sd, p + "transformer_blocks.0.", t, ctx, heads)
    if FLAGS["L8"] and t.shape[-1] % 128 == 0:
        y = _lin8(sd, p + "proj_out.", t, keep32=True).reshape(b, h, w, c).permute(0, 3, 1, 2) + x
    else:
        y = F.linear(_q("A", t), _q("W", sd[p + "proj_out.weight"]), sd[p + "proj_out.bias"]).reshape(b, h, w, c).permute(0, 3, 1, 2) + x
    return y if FLAGS["R32"] else _q("O", y)


def unet(sd, x, ts, ctx, cfg):
    FLAGS["N_off"] = False
    p = "model.diffusion_model."
    emb = nets.timestep_embedding(ts, cfg.model_channels)
    emb = F.linear(F.silu(F.linear(emb, sd[p + "time_embed.0.weight"], sd[p + "time_embed.0.bias"])), sd[p + "time_embed.2.weight"], sd[p + "time_embed.2.bias"])
    inp, mid, outp = spec.unet_schedule(cfg)

    def run(rel, layers, h):
        for j, layer in enumerate(layers):
            q = f"{p}{rel}{j}."
            kind = layer[0]
            if kind == "conv":
                h = _conv(sd, q, h)
            elif kind == "res":
                h = _resblock(sd, q, h, emb)
            elif kind == "st":
                h = _spatial_transformer(sd, q, h, ctx, layer[2])
            elif kind == "down":
                h = _conv(sd, q + "op.", h, stride=2)
            elif kind == "up":
                h = _conv(sd, q + "conv.", F.interpolate(h, scale_factor=2, mode="nearest"))
        return h
    hs = []
    h = _q("A", x)
    for i, layers in enumerate(inp):
        h = run(f"input_blocks.{i}.", layers, h)
        hs.append(h)
    h = run("middle_block.", mid, h)
    for i, layers in enumerate(outp):
        h = run(f"output_blocks.{i}.", layers, torch.cat([h, hs.pop()], dim=1))
    h = _q("N", F.silu(_gn(sd, p + "out.0.", h, 1e-5)))
    return F.conv2d(_q("A", h), _q("W", sd[p + "out.2.weight"]), sd[p + "out.2.bias"], padding=1)


def main():
    torch.set_grad_enabled(False)
    cfg = spec.EngineConfig()
    eg = np.load(os.path.join(ROOT, "tests", "golden", "engine_golden.npz"))
    sd = synth.synthetic_state_dict([(k, s) for k, s in spec.engine_param_shapes(cfg) if k.startswith("model.") or k.startswith("conditioner.embedders.0.")])
    sd["conditioner.embedders.0.pos_embedding.pe"] = nets.positional_encoding(12, 2048)
    x7 = torch.from_numpy(eg["g7_x"])
    xin = torch.cat([torch.cat([x7, x7]), torch.cat([torch.from_numpy(eg["g6_uc_concat"]), torch.from_numpy(eg["g6_c_concat"])])], dim=1)
    ctx = nets.label_encoder(sd, ["TEXT"], cfg.label)
    tctx = torch.cat([torch.zeros_like(ctx), ctx])
    ts = torch.tensor([999, 999])
    rel = lambda a, b: ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
    base = dict(W=False, A=False, O=False, N=False, N_ln=True, P=False, R32=False, A8=False, L8=False)
    FLAGS.update(base)
    ref = unet(sd, xin, ts, tctx, cfg.unet)
    print(f"injected-rounding harness vs oracle.nets.unet_forward (must be ~0): {rel(ref, nets.unet_forward(sd, xin, ts, tctx, cfg.unet)):.2e}")
    print(f"fp32 harness vs the reference golden g7_eps: {rel(ref, torch.from_numpy(eg['g7_eps'])):.2e}")
    rows = [("W   weights -> bf16", dict(W=True)), ("A   conv / linear activation operands -> bf16", dict(A=True)),
            ("O   conv / linear outputs (residual stream, intermediates) -> bf16", dict(O=True)),
            ("N   GroupNorm(+SiLU) outputs -> bf16", dict(N=True)), ("P   attention probabilities / output -> bf16", dict(P=True)),
            ("W+A (what the MFMA sees)", dict(W=True, A=True)), ("O+N (what is stored)", dict(O=True, N=True)),
            ("all = the HIP path's roundings (LayerNorms folded: never stored)", dict(W=True, A=True, O=True, N=True, P=True)),
            ("all, LayerNorm outputs ALSO stored as bf16 (UDT_LN_GEMM=0)", dict(W=True, A=True, O=True, N=True, P=True, N_ln=False)),
            ("all, fp32 residual stream (R32)", dict(W=True, A=True, O=True, N=True, P=True, R32=True)),
            ("A8  an e4m3 self-attention alone (q, k MX8 along d; v, P fixed-scale e4m3), everything else fp32", dict(A8=True)),
            ("all + A8", dict(W=True, A=True, O=True, N=True, P=True, A8=True)),
            ("L8  config #5's MX8 linears alone (640- / 1280-channel blocks), everything else fp32", dict(L8=True)),
            ("L8 + A8 (config #5's quantised arithmetic alone)", dict(L8=True, A8=True)),
            ("all + L8 = config #5 with the bf16 attention kept (UDT_FP8_ATTN=0)", dict(W=True, A=True, O=True, N=True, P=True, L8=True)),
            ("all + L8 + A8 = config #5", dict(W=True, A=True, O=True, N=True, P=True, L8=True, A8=True))]
    for name, fl in rows:
        FLAGS.update(base)
        FLAGS.update(fl)
        got = unet(sd, xin, ts, tctx, cfg.unet)
        print(f"{name:70s} rel RMS vs fp32 {rel(got, ref):.2e}")


if __name__ == "__main__":
    main()
