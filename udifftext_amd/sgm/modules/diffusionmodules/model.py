"""AutoencoderKL encoder / decoder networks on the gfx950 kernels (bf16 NHWC activations).

Module tree and state-dict names follow reference sgm/modules/diffusionmodules/model.py:
  ResnetBlock :91-148       GN(32, eps 1e-6)+swish -> conv3x3 -> GN+swish -> conv3x3 (+ 1x1 nin_shortcut), residual
                            add fused in the second conv's epilogue (temb_channels = 0 on this path)
  Downsample  :71-88        zero pad right/bottom then conv3x3 stride 2 pad 0 == gather with pad_t = pad_l = 0
  Upsample    :55-68        nearest x2 folded into the conv gather
  MemoryEfficientAttnBlock :201-262   single head, head_dim = C = 512: one q|k|v GEMM, the head_dim-512 flash kernel
                            (udt_attn512_fwd), proj_out with fused residual
  Encoder :482-596 / Decoder :599-743
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from udifftext_amd import ops, packing

from .. import hipnn as H


def Normalize(in_channels, num_groups=32):
    return H.GroupNorm(num_groups, in_channels, eps=1e-6)


class Upsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        assert with_conv
        self.conv = H.Conv2d(in_channels, in_channels, 3, padding=1)
        self.conv.emit_colstats = True            # feeds the next ResnetBlock's GroupNorm

    def forward(self, x):
        return self.conv(x, upsample=True)


class Downsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        assert with_conv
        self.conv = H.Conv2d(in_channels, in_channels, 3, stride=2, padding=0)
        self.conv.emit_colstats = True

    def forward(self, x):
        Hh, Ww = x.shape[1], x.shape[2]
        # F.pad(x, (0,1,0,1)) + conv(stride 2, pad 0): taps past the bottom/right edge read zeros
        return self.conv(x, pad=(0, 0), out_hw=((Hh + 1 - 3) // 2 + 1, (Ww + 1 - 3) // 2 + 1))


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512):
        super().__init__()
        if conv_shortcut or temb_channels > 0:
            raise NotImplementedError("VAE blocks run without time embedding / conv shortcut on this path")
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = Normalize(in_channels)
        self.conv1 = H.Conv2d(in_channels, out_channels, 3, padding=1)
        self.norm2 = Normalize(out_channels)
        self.conv2 = H.Conv2d(out_channels, out_channels, 3, padding=1)
        if in_channels != out_channels:
            self.nin_shortcut = H.Conv2d(in_channels, out_channels, 1)

    def forward(self, x, temb=None):
        # GroupNorm(eps 1e-6) -> swish -> conv3x3, twice (reference :128-143): the norms run on the convolutions' staged
        # input patches with statistics from the producers' epilogues (hipnn.Conv2d.forward, norm=)
        h = self.conv1(x, norm=self.norm1, norm_silu=True, colstats=True)
        skip = self.nin_shortcut(x) if self.in_channels != self.out_channels else x
        return self.conv2(h, residual=skip, norm=self.norm2, norm_silu=True, colstats=True)


# The VAE mid-block attention (one head of 512 dims) runs on the flash kernel (udt_attn512_fwd; a batch of 4 at 512 x 512: 256
# workgroups of 64 queries, 282 us vs 337 us for GEMM -> softmax -> GEMM in query blocks).  Grids that leave CUs idle — a single
# image: 64 workgroups, each walking all 4096 keys alone, 222 us — go through its key-split form (udt_attn512_split_fwd, round 4:
# 84 us vs 156 us for the block form; one 768 x 768 image 366 vs 607 us; profiles/r04_attn512.txt).  UDT_ATTN512=0: block form (A/B).
ATTN_FLASH_512 = os.environ.get("UDT_ATTN512", "1")


def _flash512(batch: int, n: int) -> bool:
    return ATTN_FLASH_512 != "0"


class MemoryEfficientAttnBlock(H._Packed):
    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = H.Conv2d(in_channels, in_channels, 1)
        self.k = H.Conv2d(in_channels, in_channels, 1)
        self.v = H.Conv2d(in_channels, in_channels, 1)
        self.proj_out = H.Conv2d(in_channels, in_channels, 1)

    def _key(self):
        return self.q._key() + self.k._key() + self.v._key() + self.proj_out._key()

    def fused_children(self):
        return [self.q, self.k, self.v, self.proj_out]

    def _pack(self):
        c = self.in_channels
        f = lambda m: m.weight.reshape(c, c)
        # (the biases are COPIED: `.float()` of an fp32 parameter is the parameter itself, and prepare(free_masters=True)
        #  empties the parameters — an aliased bias silently became an empty tensor; found by the GPU checkpoint test)
        # one q|k|v projection (V row-major) for the flash kernel AND the separate V^T projection of the block form
        return (H.fuse_rows(f(self.q), f(self.k), f(self.v)), torch.cat([self.q.bias, self.k.bias, self.v.bias]).float().contiguous(),
                packing.pack_linear(f(self.v)), packing.pad_bias(self.v.bias),
                packing.pack_linear(f(self.proj_out)), packing.pad_bias(self.proj_out.bias))

    def forward(self, x, **kwargs):
        B, Hh, Ww, C = x.shape
        N = Hh * Ww
        hn = self.norm(x).reshape(B * N, C)
        wqkv, bqkv, wv, bv, wo, bo = self.packed()
        if C == 512 and _flash512(B, N):
            # one q|k|v projection, then the head_dim-512 flash kernel (udt_attn512_fwd): the score tile never leaves the CU
            qkv = ops.linear(hn, wqkv, bqkv).reshape(B, N, 3 * C)
            o = ops.attention_d512(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], C ** -0.5)
        elif C == 64:
            # one head of 64 dims (not a width of the UDiffText autoencoder; the reference-module goldens use it): the UNet's flash kernel
            qkv = ops.linear(hn, wqkv, bqkv).reshape(B, N, 3 * C)
            o = ops.attention_rowv(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], 1, C ** -0.5)
        else:
            qk = ops.linear(hn, wqkv[:2 * C], bqkv[:2 * C]).reshape(B, N, 2 * C)
            vt = ops.linear(hn, wv, bv, flags=H.GEMM_TRANSPOSED, rows_per_batch=N)       # [B, C, N]
            # small grids / other widths: the scores are materialised in QUERY BLOCKS of <= 1024 rows (GEMM -> row softmax ->
            # GEMM per block): [B, 1024, N] bf16 instead of a [B, N, N] tensor
            o = torch.empty((B, N, C), dtype=torch.bfloat16, device=x.device)
            QB = 1024
            for q0 in range(0, N, QB):
                q1 = min(N, q0 + QB)
                s = ops.bmm_nt(qk[:, q0:q1, :C], qk[..., C:], alpha=C ** -0.5)           # [B, q1 - q0, N]
                ops.softmax_rows_(s)
                ops.bmm_nt(s, vt, out=o[:, q0:q1])
        out = ops.linear(o.reshape(B * N, C), wo, bo, residual=x.reshape(B * N, C), rows_per_batch=N, colstats=H.want_stats(B, N, C))
        return H.carry_stats(out.reshape(B, Hh, Ww, C), out)


def make_attn(in_channels, attn_type="vanilla", attn_kwargs=None):
    if attn_type in ("vanilla", "vanilla-xformers"):
        return MemoryEfficientAttnBlock(in_channels)
    if attn_type == "none":
        return nn.Identity()
    raise NotImplementedError(f"attn_type {attn_type} is not used by configs/test/textdesign_sd_2.yaml")


class _Level(nn.Module):
    pass


class Encoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        if len(attn_resolutions) or use_linear_attn:
            raise NotImplementedError("level attention is not configured (attn_resolutions: [])")
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.conv_in = H.Conv2d(in_channels, ch, 3, padding=1)
        self.conv_in.emit_colstats = True
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for lv in range(self.num_resolutions):
            level = _Level()
            level.block = nn.ModuleList()
            level.attn = nn.ModuleList()
            block_in, block_out = ch * in_ch_mult[lv], ch * ch_mult[lv]
            for _ in range(num_res_blocks):
                level.block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
            if lv != self.num_resolutions - 1:
                level.downsample = Downsample(block_in, resamp_with_conv)
            self.down.append(level)
        self.mid = _Level()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = H.Conv2d(block_in, 2 * z_channels if double_z else z_channels, 3, padding=1)

    def forward(self, x):
        """x: bf16 NHWC image [B, H, W, 64] (3 real channels) -> fp32 NHWC [B, H/8, W/8, 2*z]"""
        h = self.conv_in(x)
        for lv in range(self.num_resolutions):
            for blk in self.down[lv].block:
                h = blk(h)
            if lv != self.num_resolutions - 1:
                h = self.down[lv].downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        return self.conv_out(h, norm=self.norm_out, norm_silu=True)


class Decoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", **ignorekwargs):
        super().__init__()
        if len(attn_resolutions) or use_linear_attn or give_pre_end or tanh_out:
            raise NotImplementedError("decoder option not used by configs/test/textdesign_sd_2.yaml")
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.resolution, self.in_channels, self.out_ch = resolution, in_channels, out_ch
        block_in = ch * ch_mult[self.num_resolutions - 1]
        self.conv_in = H.Conv2d(z_channels, block_in, 3, padding=1)
        self.conv_in.emit_colstats = True
        self.mid = _Level()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.up = nn.ModuleList()
        for lv in reversed(range(self.num_resolutions)):
            level = _Level()
            level.block = nn.ModuleList()
            level.attn = nn.ModuleList()
            block_out = ch * ch_mult[lv]
            for _ in range(num_res_blocks + 1):
                level.block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
            if lv != 0:
                level.upsample = Upsample(block_in, resamp_with_conv)
            self.up.insert(0, level)
        self.norm_out = Normalize(block_in)
        self.conv_out = H.Conv2d(block_in, out_ch, 3, padding=1)

    def get_last_layer(self, **kwargs):
        return self.conv_out.weight

    def forward(self, z, **kwargs):
        """z: bf16 NHWC [B, h, w, 64] (z_channels real) -> fp32 NHWC [B, 8h, 8w, 4] (out_ch real channels)"""
        h = self.conv_in(z)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        for lv in reversed(range(self.num_resolutions)):
            for blk in self.up[lv].block:
                h = blk(h)
            if lv != 0:
                h = self.up[lv].upsample(h)
        return self.conv_out(h, norm=self.norm_out, norm_silu=True, flags=H.GEMM_OUT_F32)
