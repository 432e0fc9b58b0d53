"""Functional fp32 restatement of the networks on the hot path (TEST INFRASTRUCTURE, see oracle/__init__.py).

  unet_forward            UnifiedUNetModel.forward            sgm/modules/diffusionmodules/openaimodel.py:593-624
  _resblock               ResBlock.forward                    openaimodel.py:242-268
  _spatial_transformer    SpatialTransformer.forward          sgm/modules/attention.py:398-415
  _transformer_block      BasicTransformerBlock.forward       attention.py:314-341
  _self_attention         MemoryEfficientCrossAttention       attention.py:202-262 (xformers == exact softmax attention)
  _text_cross_attention   CrossAttention.forward              attention.py:140-174
  _feed_forward           FeedForward / GEGLU                 attention.py:44-70
  timestep_embedding      diffusionmodules/util.py:206-230
  vae_encode_moments      Encoder.forward + quant_conv        diffusionmodules/model.py:571-596, models/autoencoder.py:304-311
  vae_decode              post_quant_conv + Decoder.forward   models/autoencoder.py:313-316, model.py:710-743
  label_encoder           LabelEncoder.forward                sgm/modules/encoders/modules.py:1149-1173
"""
from __future__ import annotations

import math
import string
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from .spec import LabelEncoderConfig, UNetConfig, VAEConfig, unet_schedule

SD = Dict[str, torch.Tensor]


# ---------------------------------------------------------------------------------------------- UNet
def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half).to(t.device)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _gn(sd: SD, p: str, x: torch.Tensor, eps: float) -> torch.Tensor:
    return F.group_norm(x.float(), 32, sd[p + "weight"], sd[p + "bias"], eps)


def _conv(sd: SD, p: str, x: torch.Tensor, stride: int = 1, padding: int = 1) -> torch.Tensor:
    return F.conv2d(x, sd[p + "weight"], sd[p + "bias"], stride=stride, padding=padding)


def _lin(sd: SD, p: str, x: torch.Tensor, bias: bool = True) -> torch.Tensor:
    return F.linear(x, sd[p + "weight"], sd[p + "bias"] if bias else None)


def _resblock(sd: SD, p: str, x: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    h = _conv(sd, p + "in_layers.2.", F.silu(_gn(sd, p + "in_layers.0.", x, 1e-5)))
    h = h + _lin(sd, p + "emb_layers.1.", F.silu(emb))[:, :, None, None]
    h = _conv(sd, p + "out_layers.3.", F.silu(_gn(sd, p + "out_layers.0.", h, 1e-5)))
    if (p + "skip_connection.weight") in sd:
        x = _conv(sd, p + "skip_connection.", x, padding=0)
    return x + h


def _split_heads(t: torch.Tensor, heads: int) -> torch.Tensor:
    b, n, c = t.shape
    return t.reshape(b, n, heads, c // heads).permute(0, 2, 1, 3)        # b h n d


def _merge_heads(t: torch.Tensor) -> torch.Tensor:
    b, h, n, d = t.shape
    return t.permute(0, 2, 1, 3).reshape(b, n, h * d)


def _self_attention(sd: SD, p: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    q, k, v = (_split_heads(_lin(sd, p + n, x, bias=False), heads) for n in ("to_q.", "to_k.", "to_v."))
    d = q.shape[-1]
    attn = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, dim=-1)
    return _lin(sd, p + "to_out.0.", _merge_heads(attn @ v))


def _text_cross_attention(sd: SD, p: str, x: torch.Tensor, ctx: torch.Tensor, heads: int, name: str,
                          attn_maps: Optional[List[dict]]) -> torch.Tensor:
    q = _split_heads(_lin(sd, p + "to_q.", x, bias=False), heads)
    k = _split_heads(_lin(sd, p + "to_k.", ctx, bias=False), heads)
    v = _split_heads(_lin(sd, p + "to_v.", ctx, bias=False), heads)
    d = q.shape[-1]
    sim = q @ k.transpose(-1, -2) * d ** -0.5
    sim = sim.softmax(dim=-1) if sim.shape[-1] > 1 else sim.sigmoid()      # attention.py:159-162
    if attn_maps is not None:                                             # attention.py:165-169, "(b h) n l"
        b, h, n, l = sim.shape
        attn_maps.append({"name": name, "heads": heads, "size": int(n ** 0.5), "attn_map": sim.reshape(b * h, n, l)})
    return _lin(sd, p + "to_out.0.", _merge_heads(sim @ v))


def _feed_forward(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    val, gate = _lin(sd, p + "net.0.proj.", x).chunk(2, dim=-1)
    return _lin(sd, p + "net.2.", val * F.gelu(gate))


def _transformer_block(sd: SD, p: str, x: torch.Tensor, ctx: torch.Tensor, heads: int, name: str, attn_maps) -> torch.Tensor:
    c = x.shape[-1]
    ln = lambda n, t: F.layer_norm(t, (c,), sd[p + n + ".weight"], sd[p + n + ".bias"], 1e-5)
    x = _self_attention(sd, p + "attn1.", ln("norm1", x), heads) + x
    x = _text_cross_attention(sd, p + "t_attn.", ln("t_norm", x), ctx, heads, name + "t_attn", attn_maps) + x
    x = _feed_forward(sd, p + "ff.", ln("norm3", x)) + x
    return x


def _spatial_transformer(sd: SD, p: str, x: torch.Tensor, ctx: torch.Tensor, heads: int, name: str, attn_maps) -> torch.Tensor:
    b, c, h, w = x.shape
    t = _gn(sd, p + "norm.", x, 1e-6).permute(0, 2, 3, 1).reshape(b, h * w, c)
    t = _lin(sd, p + "proj_in.", t)
    t = _transformer_block(sd, p + "transformer_blocks.0.", t, ctx, heads, name + "transformer_blocks.0.", attn_maps)
    t = _lin(sd, p + "proj_out.", t)
    return t.reshape(b, h, w, c).permute(0, 3, 1, 2) + x


def _run_block(sd: SD, p: str, rel: str, layers: list, h: torch.Tensor, emb: torch.Tensor, ctx: torch.Tensor, attn_maps):
    for j, layer in enumerate(layers):
        q = f"{p}{rel}{j}."
        kind = layer[0]
        if kind == "conv":
            h = _conv(sd, q, h)
        elif kind == "res":
            h = _resblock(sd, q, h, emb)
        elif kind == "st":
            h = _spatial_transformer(sd, q, h, ctx, layer[2], f"{rel}{j}.", attn_maps)
        elif kind == "down":
            h = _conv(sd, q + "op.", h, stride=2)
        elif kind == "up":
            h = _conv(sd, q + "conv.", F.interpolate(h, scale_factor=2, mode="nearest"))
    return h


def unet_forward(sd: SD, x: torch.Tensor, timesteps: torch.Tensor, t_context: torch.Tensor,
                 cfg: Optional[UNetConfig] = None, prefix: str = "model.diffusion_model.",
                 attn_maps: Optional[List[dict]] = None, taps: Optional[dict] = None) -> torch.Tensor:
    """x [B, 9, h, w], timesteps [B], t_context [B, L, 2048] -> eps [B, 4, h, w].
    ``attn_maps`` (list) receives the t_attn probability maps in module order; ``taps`` (dict) receives the
    activation after every block (for per-block parity checks)."""
    cfg = cfg or UNetConfig()
    inputs, middle, outputs = unet_schedule(cfg)
    emb = timestep_embedding(timesteps, cfg.model_channels)
    emb = _lin(sd, prefix + "time_embed.2.", F.silu(_lin(sd, prefix + "time_embed.0.", emb)))
    hs = []
    h = x
    for i, layers in enumerate(inputs):
        h = _run_block(sd, prefix, f"input_blocks.{i}.", layers, h, emb, t_context, attn_maps)
        hs.append(h)
        if taps is not None:
            taps[f"input_blocks.{i}"] = h
    h = _run_block(sd, prefix, "middle_block.", middle, h, emb, t_context, attn_maps)
    if taps is not None:
        taps["middle_block"] = h
    for i, layers in enumerate(outputs):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_block(sd, prefix, f"output_blocks.{i}.", layers, h, emb, t_context, attn_maps)
        if taps is not None:
            taps[f"output_blocks.{i}"] = h
    return _conv(sd, prefix + "out.2.", F.silu(_gn(sd, prefix + "out.0.", h, 1e-5)))


# ----------------------------------------------------------------------------------------------- VAE
def _vae_resnet(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    h = _conv(sd, p + "conv1.", F.silu(_gn(sd, p + "norm1.", x, 1e-6)))
    h = _conv(sd, p + "conv2.", F.silu(_gn(sd, p + "norm2.", h, 1e-6)))
    if (p + "nin_shortcut.weight") in sd:
        x = _conv(sd, p + "nin_shortcut.", x, padding=0)
    return x + h


def _vae_attn(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    b, c, h, w = x.shape
    t = _gn(sd, p + "norm.", x, 1e-6)
    q, k, v = (_conv(sd, p + n, t, padding=0).reshape(b, c, h * w).transpose(1, 2) for n in ("q.", "k.", "v."))
    attn = torch.softmax(q @ k.transpose(-1, -2) * c ** -0.5, dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(b, c, h, w)
    return x + _conv(sd, p + "proj_out.", o, padding=0)


def vae_encode_moments(sd: SD, x: torch.Tensor, cfg: Optional[VAEConfig] = None, prefix: str = "first_stage_model.") -> torch.Tensor:
    """image [B,3,H,W] -> moments [B,8,H/8,W/8] (mean ‖ logvar)."""
    cfg = cfg or VAEConfig()
    e = prefix + "encoder."
    h = _conv(sd, e + "conv_in.", x)
    nlev = len(cfg.ch_mult)
    for lv in range(nlev):
        for b in range(cfg.num_res_blocks):
            h = _vae_resnet(sd, f"{e}down.{lv}.block.{b}.", h)
        if lv != nlev - 1:
            h = _conv(sd, f"{e}down.{lv}.downsample.conv.", F.pad(h, (0, 1, 0, 1)), stride=2, padding=0)   # model.py:81-85
    h = _vae_resnet(sd, e + "mid.block_1.", h)
    h = _vae_attn(sd, e + "mid.attn_1.", h)
    h = _vae_resnet(sd, e + "mid.block_2.", h)
    h = _conv(sd, e + "conv_out.", F.silu(_gn(sd, e + "norm_out.", h, 1e-6)))
    return _conv(sd, prefix + "quant_conv.", h, padding=0)


def posterior_sample(moments: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """DiagonalGaussianDistribution.sample, distributions.py:24-41 (noise drawn by the caller, on the CPU)."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return mean + torch.exp(0.5 * logvar) * noise


def vae_decode(sd: SD, z: torch.Tensor, cfg: Optional[VAEConfig] = None, prefix: str = "first_stage_model.") -> torch.Tensor:
    """latent [B,4,h,w] (already divided by scale_factor) -> image [B,3,8h,8w]."""
    cfg = cfg or VAEConfig()
    d = prefix + "decoder."
    h = _conv(sd, d + "conv_in.", _conv(sd, prefix + "post_quant_conv.", z, padding=0))
    h = _vae_resnet(sd, d + "mid.block_1.", h)
    h = _vae_attn(sd, d + "mid.attn_1.", h)
    h = _vae_resnet(sd, d + "mid.block_2.", h)
    for lv in reversed(range(len(cfg.ch_mult))):
        for b in range(cfg.num_res_blocks + 1):
            h = _vae_resnet(sd, f"{d}up.{lv}.block.{b}.", h)
        if lv != 0:
            h = _conv(sd, f"{d}up.{lv}.upsample.conv.", F.interpolate(h, scale_factor=2.0, mode="nearest"))
    return _conv(sd, d + "conv_out.", F.silu(_gn(sd, d + "norm_out.", h, 1e-6)))


# -------------------------------------------------------------------------------------- LabelEncoder
CHARSET = string.printable[:-6]


def label_indices(labels: List[str], max_len: int = 12) -> torch.Tensor:
    """chars -> 1-based index into string.printable[:-6]; unknown chars and padding -> 0 (modules.py:1149-1158)."""
    rows = []
    for s in labels:
        if len(s) > max_len:
            raise AssertionError("label longer than max_len")
        idx = [CHARSET.find(c) + 1 for c in s]
        rows.append(idx + [0] * (max_len - len(idx)))
    return torch.tensor(rows, dtype=torch.long)


def positional_encoding(max_len: int, d_model: int) -> torch.Tensor:
    """PositionalEncoding.pe, modules.py:1076-1081."""
    pe = torch.zeros(max_len, d_model)
    pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def label_encoder(sd: SD, labels: List[str], cfg: Optional[LabelEncoderConfig] = None,
                  prefix: str = "conditioner.embedders.0.") -> torch.Tensor:
    """List[str] -> [B, max_len, emb_dim]; 12 post-norm encoder layers, ReLU FF, NO padding mask."""
    cfg = cfg or LabelEncoderConfig()
    idx = label_indices(labels, cfg.max_len)
    d, H = cfg.emb_dim, cfg.n_heads
    x = sd[prefix + "label_embedding.weight"][idx] + sd[prefix + "pos_embedding.pe"][None]
    for i in range(cfg.n_layers):
        p = f"{prefix}encoder.layers.{i}."
        qkv = F.linear(x, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"])
        q, k, v = (_split_heads(t, H) for t in qkv.chunk(3, dim=-1))
        attn = torch.softmax(q @ k.transpose(-1, -2) * (d // H) ** -0.5, dim=-1)
        sa = _lin(sd, p + "self_attn.out_proj.", _merge_heads(attn @ v))
        x = F.layer_norm(x + sa, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
        ff = _lin(sd, p + "linear2.", F.relu(_lin(sd, p + "linear1.", x)))
        x = F.layer_norm(x + ff, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    return x
