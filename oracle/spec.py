"""Architecture description of the UDiffText engine: block schedule + checkpoint key names / shapes.

Restates the constructors of the reference so that the oracle needs no nn.Module:
  UnifiedUNetModel.__init__   sgm/modules/diffusionmodules/openaimodel.py:275-550
  ResBlock.__init__           sgm/modules/diffusionmodules/openaimodel.py:154-240
  SpatialTransformer/BasicTransformerBlock/CrossAttention.__init__   sgm/modules/attention.py:111-138,179-200,267-312,354-396
  Encoder/Decoder/ResnetBlock/MemoryEfficientAttnBlock.__init__      sgm/modules/diffusionmodules/model.py:91-126,207-226,481-566,599-700
  AutoencoderKL.__init__      sgm/models/autoencoder.py:282-302
  LabelEncoder.__init__       sgm/modules/encoders/modules.py:1088-1107
  config values               configs/test/textdesign_sd_2.yaml

tests/golden/state_dict_keys.json (dumped from the real reference engine) pins every name and shape.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Tuple

Shape = Tuple[int, ...]


@dataclass
class UNetConfig:            # configs/test/textdesign_sd_2.yaml:22-37
    in_channels: int = 9
    out_channels: int = 4
    model_channels: int = 320
    attention_resolutions: Tuple[int, ...] = (4, 2, 1)
    num_res_blocks: int = 2
    channel_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_head_channels: int = 64
    t_context_dim: int = 2048

    @property
    def time_embed_dim(self) -> int:
        return 4 * self.model_channels


@dataclass
class VAEConfig:             # configs/test/textdesign_sd_2.yaml:72-83
    ch: int = 128
    ch_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    z_channels: int = 4
    in_channels: int = 3
    out_ch: int = 3
    embed_dim: int = 4


@dataclass
class LabelEncoderConfig:    # configs/test/textdesign_sd_2.yaml:49-55
    max_len: int = 12
    emb_dim: int = 2048
    n_heads: int = 8
    n_layers: int = 12
    num_cls: int = 95        # len(string.printable[:-6]) + 1
    ff_dim: int = 2048       # nn.TransformerEncoderLayer default dim_feedforward


@dataclass
class EngineConfig:
    unet: UNetConfig = field(default_factory=UNetConfig)
    vae: VAEConfig = field(default_factory=VAEConfig)
    label: LabelEncoderConfig = field(default_factory=LabelEncoderConfig)
    scale_factor: float = 0.18215


# ------------------------------------------------------------------------------------------- UNet schedule
def unet_schedule(cfg: UNetConfig):
    """Returns (input_blocks, middle_block, output_blocks); each block is a list of layer tuples:
       ("conv", cin, cout) | ("res", cin, cout) | ("st", ch, heads) | ("down", ch) | ("up", ch)
    following the constructor loop openaimodel.py:355-533."""
    mc = cfg.model_channels
    inputs: List[list] = [[("conv", cfg.in_channels, mc)]]
    chans = [mc]
    ch, ds = mc, 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            layers = [("res", ch, mult * mc)]
            ch = mult * mc
            if ds in cfg.attention_resolutions:
                layers.append(("st", ch, ch // cfg.num_head_channels))
            inputs.append(layers)
            chans.append(ch)
        if level != len(cfg.channel_mult) - 1:
            inputs.append([("down", ch)])
            chans.append(ch)
            ds *= 2
    heads = ch // cfg.num_head_channels
    middle = [("res", ch, ch), ("st", ch, heads), ("res", ch, ch)]
    outputs: List[list] = []
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, mc * mult)]
            ch = mc * mult
            if ds in cfg.attention_resolutions:
                layers.append(("st", ch, ch // cfg.num_head_channels))
            if level and i == cfg.num_res_blocks:
                layers.append(("up", ch))
                ds //= 2
            outputs.append(layers)
    return inputs, middle, outputs


def _res_shapes(p: str, cin: int, cout: int, temb: int) -> List[Tuple[str, Shape]]:
    s = [(p + "in_layers.0.weight", (cin,)), (p + "in_layers.0.bias", (cin,)),
         (p + "in_layers.2.weight", (cout, cin, 3, 3)), (p + "in_layers.2.bias", (cout,)),
         (p + "emb_layers.1.weight", (cout, temb)), (p + "emb_layers.1.bias", (cout,)),
         (p + "out_layers.0.weight", (cout,)), (p + "out_layers.0.bias", (cout,)),
         (p + "out_layers.3.weight", (cout, cout, 3, 3)), (p + "out_layers.3.bias", (cout,))]
    if cin != cout:
        s += [(p + "skip_connection.weight", (cout, cin, 1, 1)), (p + "skip_connection.bias", (cout,))]
    return s


def _st_shapes(p: str, ch: int, ctx: int) -> List[Tuple[str, Shape]]:
    b = p + "transformer_blocks.0."
    s = [(p + "norm.weight", (ch,)), (p + "norm.bias", (ch,)),
         (p + "proj_in.weight", (ch, ch)), (p + "proj_in.bias", (ch,))]
    for a, kdim in (("attn1", ch), ("t_attn", ctx)):
        s += [(b + a + ".to_q.weight", (ch, ch)), (b + a + ".to_k.weight", (ch, kdim)),
              (b + a + ".to_v.weight", (ch, kdim)),
              (b + a + ".to_out.0.weight", (ch, ch)), (b + a + ".to_out.0.bias", (ch,))]
    for n in ("t_norm", "norm1", "norm3"):
        s += [(b + n + ".weight", (ch,)), (b + n + ".bias", (ch,))]
    s += [(b + "ff.net.0.proj.weight", (8 * ch, ch)), (b + "ff.net.0.proj.bias", (8 * ch,)),
          (b + "ff.net.2.weight", (ch, 4 * ch)), (b + "ff.net.2.bias", (ch,)),
          (p + "proj_out.weight", (ch, ch)), (p + "proj_out.bias", (ch,))]
    return s


def _block_shapes(p: str, layers: list, cfg: UNetConfig) -> List[Tuple[str, Shape]]:
    out: List[Tuple[str, Shape]] = []
    for j, layer in enumerate(layers):
        q = f"{p}{j}."
        kind = layer[0]
        if kind == "conv":
            out += [(q + "weight", (layer[2], layer[1], 3, 3)), (q + "bias", (layer[2],))]
        elif kind == "res":
            out += _res_shapes(q, layer[1], layer[2], cfg.time_embed_dim)
        elif kind == "st":
            out += _st_shapes(q, layer[1], cfg.t_context_dim)
        elif kind == "down":
            out += [(q + "op.weight", (layer[1], layer[1], 3, 3)), (q + "op.bias", (layer[1],))]
        elif kind == "up":
            out += [(q + "conv.weight", (layer[1], layer[1], 3, 3)), (q + "conv.bias", (layer[1],))]
    return out


def unet_param_shapes(cfg: UNetConfig, prefix: str = "model.diffusion_model.") -> List[Tuple[str, Shape]]:
    mc, te = cfg.model_channels, cfg.time_embed_dim
    s = [(prefix + "time_embed.0.weight", (te, mc)), (prefix + "time_embed.0.bias", (te,)),
         (prefix + "time_embed.2.weight", (te, te)), (prefix + "time_embed.2.bias", (te,))]
    inputs, middle, outputs = unet_schedule(cfg)
    for i, layers in enumerate(inputs):
        s += _block_shapes(f"{prefix}input_blocks.{i}.", layers, cfg)
    s += _block_shapes(prefix + "middle_block.", middle, cfg)
    for i, layers in enumerate(outputs):
        s += _block_shapes(f"{prefix}output_blocks.{i}.", layers, cfg)
    s += [(prefix + "out.0.weight", (mc,)), (prefix + "out.0.bias", (mc,)),
          (prefix + "out.2.weight", (cfg.out_channels, mc, 3, 3)), (prefix + "out.2.bias", (cfg.out_channels,))]
    return s


# ------------------------------------------------------------------------------------------------ VAE
def _resnet_shapes(p: str, cin: int, cout: int) -> List[Tuple[str, Shape]]:
    s = [(p + "norm1.weight", (cin,)), (p + "norm1.bias", (cin,)),
         (p + "conv1.weight", (cout, cin, 3, 3)), (p + "conv1.bias", (cout,)),
         (p + "norm2.weight", (cout,)), (p + "norm2.bias", (cout,)),
         (p + "conv2.weight", (cout, cout, 3, 3)), (p + "conv2.bias", (cout,))]
    if cin != cout:
        s += [(p + "nin_shortcut.weight", (cout, cin, 1, 1)), (p + "nin_shortcut.bias", (cout,))]
    return s


def _vae_attn_shapes(p: str, c: int) -> List[Tuple[str, Shape]]:
    s = [(p + "norm.weight", (c,)), (p + "norm.bias", (c,))]
    for n in ("q", "k", "v", "proj_out"):
        s += [(p + n + ".weight", (c, c, 1, 1)), (p + n + ".bias", (c,))]
    return s


def vae_param_shapes(cfg: VAEConfig, prefix: str = "first_stage_model.") -> List[Tuple[str, Shape]]:
    ch, nres = cfg.ch, cfg.num_res_blocks
    nlev = len(cfg.ch_mult)
    in_mult = (1,) + tuple(cfg.ch_mult)
    s: List[Tuple[str, Shape]] = []
    # encoder (model.py:509-566)
    e = prefix + "encoder."
    s += [(e + "conv_in.weight", (ch, cfg.in_channels, 3, 3)), (e + "conv_in.bias", (ch,))]
    bi = ch
    for lv in range(nlev):
        bi, bo = ch * in_mult[lv], ch * cfg.ch_mult[lv]
        for b in range(nres):
            s += _resnet_shapes(f"{e}down.{lv}.block.{b}.", bi, bo)
            bi = bo
        if lv != nlev - 1:
            s += [(f"{e}down.{lv}.downsample.conv.weight", (bi, bi, 3, 3)), (f"{e}down.{lv}.downsample.conv.bias", (bi,))]
    s += _resnet_shapes(e + "mid.block_1.", bi, bi) + _vae_attn_shapes(e + "mid.attn_1.", bi) + \
        _resnet_shapes(e + "mid.block_2.", bi, bi)
    s += [(e + "norm_out.weight", (bi,)), (e + "norm_out.bias", (bi,)),
          (e + "conv_out.weight", (2 * cfg.z_channels, bi, 3, 3)), (e + "conv_out.bias", (2 * cfg.z_channels,))]
    # decoder (model.py:634-700); module registration order: conv_in, mid, up (inserted at 0), norm_out, conv_out
    d = prefix + "decoder."
    bi = ch * cfg.ch_mult[-1]
    s += [(d + "conv_in.weight", (bi, cfg.z_channels, 3, 3)), (d + "conv_in.bias", (bi,))]
    s += _resnet_shapes(d + "mid.block_1.", bi, bi) + _vae_attn_shapes(d + "mid.attn_1.", bi) + \
        _resnet_shapes(d + "mid.block_2.", bi, bi)
    ups: Dict[int, List[Tuple[str, Shape]]] = {}
    for lv in reversed(range(nlev)):
        bo = ch * cfg.ch_mult[lv]
        cur: List[Tuple[str, Shape]] = []
        for b in range(nres + 1):
            cur += _resnet_shapes(f"{d}up.{lv}.block.{b}.", bi, bo)
            bi = bo
        if lv != 0:
            cur += [(f"{d}up.{lv}.upsample.conv.weight", (bi, bi, 3, 3)), (f"{d}up.{lv}.upsample.conv.bias", (bi,))]
        ups[lv] = cur
    for lv in range(nlev):
        s += ups[lv]
    s += [(d + "norm_out.weight", (bi,)), (d + "norm_out.bias", (bi,)),
          (d + "conv_out.weight", (cfg.out_ch, bi, 3, 3)), (d + "conv_out.bias", (cfg.out_ch,))]
    s += [(prefix + "quant_conv.weight", (2 * cfg.embed_dim, 2 * cfg.z_channels, 1, 1)),
          (prefix + "quant_conv.bias", (2 * cfg.embed_dim,)),
          (prefix + "post_quant_conv.weight", (cfg.z_channels, cfg.embed_dim, 1, 1)),
          (prefix + "post_quant_conv.bias", (cfg.z_channels,))]
    return s


# ---------------------------------------------------------------------------------------- LabelEncoder
def label_encoder_param_shapes(cfg: LabelEncoderConfig, prefix: str = "conditioner.embedders.0.") -> List[Tuple[str, Shape]]:
    d = cfg.emb_dim
    s = [(prefix + "label_embedding.weight", (cfg.num_cls, d)), (prefix + "pos_embedding.pe", (cfg.max_len, d))]
    for i in range(cfg.n_layers):
        p = f"{prefix}encoder.layers.{i}."
        s += [(p + "self_attn.in_proj_weight", (3 * d, d)), (p + "self_attn.in_proj_bias", (3 * d,)),
              (p + "self_attn.out_proj.weight", (d, d)), (p + "self_attn.out_proj.bias", (d,)),
              (p + "linear1.weight", (cfg.ff_dim, d)), (p + "linear1.bias", (cfg.ff_dim,)),
              (p + "linear2.weight", (d, cfg.ff_dim)), (p + "linear2.bias", (d,)),
              (p + "norm1.weight", (d,)), (p + "norm1.bias", (d,)),
              (p + "norm2.weight", (d,)), (p + "norm2.bias", (d,))]
    return s


def engine_param_shapes(cfg: EngineConfig) -> List[Tuple[str, Shape]]:
    """All 1330 state-dict entries of the DiffusionEngine (SURVEY.md §5 'Checkpoint / resume')."""
    s = unet_param_shapes(cfg.unet)
    s += [("denoiser.sigmas", (1000,))]
    s += label_encoder_param_shapes(cfg.label)
    s += vae_param_shapes(cfg.vae, "conditioner.embedders.2.model.")
    s += vae_param_shapes(cfg.vae, "first_stage_model.")
    s += [("loss_fn.g_kernel", (cfg.label.max_len, 1, 3, 3))]
    return s
