"""The SD-2-inpainting UNet (``UnifiedUNetModel``) on the gfx950 kernels.

Constructor arguments, module tree and state-dict names follow reference
sgm/modules/diffusionmodules/openaimodel.py:275-550; ``forward`` keeps the reference signature
(:593-624: NCHW fp32 in, NCHW fp32 eps out, ``attn_map_cache`` filled with the t_attn probabilities).
Internally activations are bf16 channel-last and every op is a launch into libudt_kernels.so:

  ResBlock (:149-268)   GN32+SiLU -> conv3x3 (+time-embedding row vector in the epilogue)
                        -> GN32+SiLU -> conv3x3 (+skip / 1x1-conv skip in the epilogue)
  skip concat (:620)    never materialised for the convolutions: the GroupNorm kernels and the 1x1 skip conv read
                        the two sources directly
  Upsample (:89-102)    nearest x2 folded into the conv's gather;  Downsample (:144-146) conv3x3 stride 2
  emb_layers (:210-216) all 22 Linear(SiLU(emb)) evaluated by ONE GEMM per call
"""
from __future__ import annotations

import os

from typing import List, Optional

import torch
import torch.nn as nn

from udifftext_amd import ops, packing

from ...util import default, exists, require_gpu
from .. import hipnn as H
from ..attention import SpatialTransformer
from .util import zero_module

CPAD = packing.KPAD


class Timestep(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, t):
        return ops.timestep_embedding(t.float().contiguous(), self.dim)


class TimestepBlock(nn.Module):
    """a module whose forward takes the time-embedding contribution as second argument"""


class Upsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1, third_up=False):
        super().__init__()
        assert dims == 2 and use_conv
        self.channels = channels
        self.out_channels = out_channels or channels
        self.conv = H.Conv2d(channels, self.out_channels, 3, padding=padding)
        self.conv.emit_colstats = True            # feeds the next ResBlock's GroupNorm

    def forward(self, x):
        assert x.shape[-1] == self.channels
        return self.conv(x, upsample=True)


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1, third_down=False):
        super().__init__()
        assert dims == 2 and use_conv
        self.channels = channels
        self.out_channels = out_channels or channels
        self.op = H.Conv2d(channels, self.out_channels, 3, stride=2, padding=padding)
        self.op.emit_colstats = True              # feeds the next ResBlock's GroupNorm (and a skip concat later)

    def forward(self, x):
        assert x.shape[-1] == self.channels
        return self.op(x)


class ResBlock(TimestepBlock):
    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False, use_scale_shift_norm=False,
                 dims=2, up=False, down=False, kernel_size=3, exchange_temb_dims=False, skip_t_emb=False):
        super().__init__()
        if use_scale_shift_norm or up or down or use_conv or skip_t_emb or dims != 2 or kernel_size != 3:
            raise NotImplementedError("ResBlock variant not used by configs/test/textdesign_sd_2.yaml")
        self.channels = channels
        self.emb_channels = emb_channels
        self.out_channels = out_channels or channels
        self.in_layers = nn.Sequential(H.GroupNorm(32, channels), nn.Identity(),
                                       H.Conv2d(channels, self.out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.Identity(), H.Linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(H.GroupNorm(32, self.out_channels), nn.Identity(), nn.Identity(),
                                        zero_module(H.Conv2d(self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = H.Conv2d(channels, self.out_channels, 1)

    def forward(self, x, emb_rows, x2=None):
        """x (+ optional skip source x2, concatenated on channels): bf16 NHWC; emb_rows: fp32 [B, *] whose columns
        [emb_offset, emb_offset + out_channels) hold this block's Linear(SiLU(emb)) (UnifiedUNetModel.time_embedding_rows)."""
        emb_out = emb_rows[:, self.emb_offset:self.emb_offset + self.out_channels]
        # GroupNorm32 -> SiLU -> conv3x3 twice (reference :183-187,218-231): the norms run on the convolutions' staged
        # input patches, their statistics come out of the producers' epilogues (hipnn.Conv2d.forward, norm=)
        h = self.in_layers[2](x, x2=x2, rowvec=emb_out, norm=self.in_layers[0], norm_silu=True, colstats=True)
        if isinstance(self.skip_connection, nn.Identity):
            assert x2 is None
            skip = x
        else:
            skip = self.skip_connection(x, x2=x2)
        return self.out_layers[3](h, residual=skip, norm=self.out_layers[0], norm_silu=True, colstats=True)


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    def forward(self, x, emb_rows=None, t_context=None, v_context=None, x2=None, t_kv=None, emit_map=False,
                zero_ctx_rows=0, t_fused=None):
        for layer in self:
            if isinstance(layer, ResBlock):
                x = layer(x, emb_rows, x2=x2)
                x2 = None
            elif isinstance(layer, SpatialTransformer):
                kv = t_kv[layer.st_index] if t_kv is not None else None
                tf = t_fused[layer.st_index] if t_fused is not None else None
                x = layer(x, t_context, v_context, t_kv=kv, emit_map=emit_map, zero_ctx_rows=zero_ctx_rows, t_fused=tf)
            else:
                x = layer(x)
        return x


class UnifiedUNetModel(nn.Module):
    def __init__(self, in_channels, ctrl_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), save_attn_type=None, save_attn_layers=[], conv_resample=True,
                 dims=2, use_label=None, num_heads=-1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False, transformer_depth=1, t_context_dim=None,
                 v_context_dim=None, num_attention_blocks=None, use_linear_in_transformer=False, adm_in_channels=None,
                 transformer_depth_middle=None):
        super().__init__()
        if ctrl_channels or use_label is not None or resblock_updown or dims != 2 or not conv_resample:
            raise NotImplementedError("option not used by configs/test/textdesign_sd_2.yaml")
        if num_head_channels == -1:
            assert num_heads != -1, "Either num_heads or num_head_channels has to be set"
        self.in_channels, self.ctrl_channels = in_channels, ctrl_channels
        self.model_channels, self.out_channels = model_channels, out_channels
        channel_mult = tuple(channel_mult)
        depth = len(channel_mult) * [transformer_depth]
        depth_mid = default(transformer_depth_middle, depth[-1])
        self.num_res_blocks = len(channel_mult) * [num_res_blocks]
        self.attention_resolutions = tuple(attention_resolutions)
        self.channel_mult = channel_mult
        self.use_label = use_label
        self.num_heads, self.num_head_channels = num_heads, num_head_channels

        ted = model_channels * 4
        self.time_embed = nn.Sequential(H.Linear(model_channels, ted), nn.Identity(), H.Linear(ted, ted))

        def heads_of(ch):
            return (num_heads, ch // num_heads) if num_head_channels == -1 else (ch // num_head_channels, num_head_channels)

        def transformer(ch, d):
            nh, dh = heads_of(ch)
            return SpatialTransformer(ch, nh, dh, depth=d, t_context_dim=t_context_dim, v_context_dim=v_context_dim,
                                      use_linear=use_linear_in_transformer)

        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(H.Conv2d(in_channels, model_channels, 3, padding=1))])
        self.input_blocks[0][0].emit_colstats = True
        chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for nr in range(self.num_res_blocks[level]):
                layers = [ResBlock(ch, ted, dropout, out_channels=mult * model_channels)]
                ch = mult * model_channels
                if ds in self.attention_resolutions and (not exists(num_attention_blocks) or nr < num_attention_blocks[level]):
                    layers.append(transformer(ch, depth[level]))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, True, out_channels=ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(ResBlock(ch, ted, dropout), transformer(ch, depth_mid),
                                                    ResBlock(ch, ted, dropout))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(self.num_res_blocks[level] + 1):
                ich = chans.pop()
                layers = [ResBlock(ch + ich, ted, dropout, out_channels=model_channels * mult)]
                ch = model_channels * mult
                if ds in self.attention_resolutions and (not exists(num_attention_blocks) or i < num_attention_blocks[level]):
                    layers.append(transformer(ch, depth[level]))
                if level and i == self.num_res_blocks[level]:
                    layers.append(Upsample(ch, True, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(H.GroupNorm(32, ch), nn.Identity(),
                                 zero_module(H.Conv2d(model_channels, out_channels, 3, padding=1)))

        # attention-map cache (reference :542-550): one dict per module whose name ends with a saved type
        self.attn_type = save_attn_type or []
        self.attn_layers = save_attn_layers
        self.attn_map_cache = []
        for name, module in self.named_modules():
            if any(name.endswith(t) for t in self.attn_type):
                item = {"name": name, "heads": module.heads, "size": None, "attn_map": None}
                self.attn_map_cache.append(item)
                module.attn_map_cache = item
        self.cache_attn_maps = True          # forward() fills attn_map_cache like the reference; the sampler's
        #                                      main loop switches it off (the maps are write-only there)
        self._resblocks = [m for m in self.modules() if isinstance(m, ResBlock)]
        self._transformers = [m for m in self.modules() if isinstance(m, SpatialTransformer)]
        off = 0
        for rb in self._resblocks:                 # column range of each block inside time_embedding_rows()
            rb.emb_offset = off
            off += rb.out_channels
        for i, st in enumerate(self._transformers):
            st.st_index = i

    # ------------------------------------------------------------------------------------------ helpers
    def clear_attn_map(self):
        for item in self.attn_map_cache:
            item["attn_map"] = None

    def save_attn_map(self, attn_type="t_attn", save_name="temp", tokens="", out_dir="temp/attn_map"):
        """reference openaimodel.py:559-591: the cached ``attn_type`` probabilities of the layers named by ``save_attn_layers``
        (configs: output_blocks.6.1), averaged over layers and heads -> per-token heat maps [L, h, w] of the LAST sample of the
        batch; a 3 x 4 grid of the first 12 is written to ``out_dir``/attn_map_<save_name>.png (matplotlib's imshow where the
        reference draws seaborn heatmaps — the returned array is what the sampler passes on to save_segment_map).  Host-side only:
        the maps are whatever the last map-emitting UNet call cached."""
        maps, heads = [], 1
        for item in self.attn_map_cache:
            name = item["name"]
            if any(name.startswith(block) for block in self.attn_layers) and name.endswith(attn_type):
                if item["attn_map"] is None:
                    raise RuntimeError(f"save_attn_map: no cached map for {name} — run a map-emitting UNet call first")
                heads = item["heads"]
                maps.append(item["attn_map"].detach().float().cpu())
        if not maps:
            raise RuntimeError("save_attn_map: save_attn_layers / save_attn_type select no attention layer")
        attn_map = torch.stack(maps, dim=0).mean(dim=0)                     # [b * heads, n, l]
        bh, n, l = attn_map.shape
        attn_map = attn_map.reshape(-1, heads, n, l).mean(dim=1)             # [b, n, l]
        b = attn_map.shape[0]
        h = w = int(n ** 0.5)
        attn_map_i = attn_map.permute(0, 2, 1).reshape(b, l, h, w).numpy()[-1]
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            os.makedirs(out_dir, exist_ok=True)
            fig = plt.figure(figsize=(12, 8), dpi=100)
            for j in range(min(12, attn_map_i.shape[0])):
                ax = fig.add_subplot(3, 4, j + 1)
                ax.imshow(attn_map_i[j])
                ax.set_xticks([]); ax.set_yticks([])
                if j < len(tokens):
                    ax.set_title(tokens[j])
            fig.savefig(os.path.join(out_dir, f"attn_map_{save_name}.png"))
            plt.close(fig)
        except ImportError:                                                  # (no plotting backend: the array is still returned)
            pass
        return attn_map_i

    def _emb_pack(self):
        if getattr(self, "_emb_frozen", False):
            return self._emb_w, self._emb_b
        key = tuple(rb.emb_layers[1]._key() for rb in self._resblocks)
        if getattr(self, "_emb_key", None) != key:
            with torch.no_grad():
                w = packing.pack_linear(torch.cat([rb.emb_layers[1].weight for rb in self._resblocks], 0))
                b = torch.cat([rb.emb_layers[1].bias for rb in self._resblocks], 0).float().contiguous()
            self._emb_w, self._emb_b, self._emb_key = w, b, key
        return self._emb_w, self._emb_b

    def time_embedding_rows(self, timesteps: torch.Tensor) -> torch.Tensor:
        """timesteps [B] -> fp32 [B, sum(out_channels of all ResBlocks)]: every block's Linear(SiLU(emb))."""
        t_emb = ops.timestep_embedding(timesteps.float().contiguous(), self.model_channels)
        e = self.time_embed[0](t_emb, flags=H.GEMM_SILU_OUT)
        se = self.time_embed[2](e, flags=H.GEMM_SILU_OUT)            # SiLU(emb): the only way emb is consumed
        w, b = self._emb_pack()
        return ops.linear(se, w, b, flags=H.GEMM_OUT_F32)

    def project_context(self, t_context: torch.Tensor) -> List[list]:
        """k|v projections of the text context for all transformers (step-invariant)"""
        ctx = t_context.to(torch.bfloat16).contiguous()
        return [st.project_context(ctx) for st in self._transformers]

    def prepare_fused_tattn(self, t_kv: List[list], out: Optional[List[list]] = None) -> List[list]:
        """fold the hoisted context projections of all transformers into the tables of the fused text cross-attention
        (udt_tattn_prepare); ``out``: refresh existing tables in place (static buffers of captured graphs)"""
        return [st.prepare_fused(kv, out=(out[i] if out is not None else None))
                for i, (st, kv) in enumerate(zip(self._transformers, t_kv))]

    def forward_nhwc(self, xin: torch.Tensor, emb_rows: torch.Tensor, t_kv: List[list], emit_maps: bool = False,
                     zero_ctx_rows: int = 0, t_fused: Optional[List[list]] = None) -> torch.Tensor:
        """xin: bf16 [B, h, w, 64] (9 real channels); returns eps fp32 [B, h, w, 4].
        zero_ctx_rows: leading samples whose text context is exactly zero (see BasicTransformerBlock.forward)."""
        hs = []
        h = xin
        kw = dict(t_kv=t_kv, emit_map=emit_maps, zero_ctx_rows=zero_ctx_rows, t_fused=t_fused)
        for block in self.input_blocks:
            h = block(h, emb_rows, **kw)
            hs.append(h)
        h = self.middle_block(h, emb_rows, **kw)
        for block in self.output_blocks:
            h = block(h, emb_rows, x2=hs.pop(), **kw)
        return self.out[2](h, norm=self.out[0], norm_silu=True, flags=H.GEMM_OUT_F32, colstats=False)

    def forward(self, x, timesteps=None, t_context=None, v_context=None, y=None, **kwargs):
        """reference signature: x [B, in_channels, h, w] fp32, timesteps [B], t_context [B, L, Dc] -> [B, out, h, w]"""
        assert (y is not None) == (self.use_label is not None), \
            "must specify y if and only if the model is class-conditional"
        require_gpu(x, "UnifiedUNetModel.forward")
        self.clear_attn_map()
        B, Cc, h, w = x.shape
        assert Cc == self.in_channels
        xin = ops.nchw_to_nhwc(x.float().contiguous(), CPAD)
        emb_rows = self.time_embedding_rows(timesteps)
        t_kv = self.project_context(t_context)
        eps = self.forward_nhwc(xin, emb_rows, t_kv, emit_maps=self.cache_attn_maps)
        return ops.nhwc_to_nchw(eps, self.out_channels).type(x.dtype)
