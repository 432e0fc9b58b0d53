"""Attend-and-excite on the CPU oracle: d local_loss / d x by torch.autograd through the functional fp32 UNet of
oracle/nets.py (TEST INFRASTRUCTURE — only tests/, never the product path).

  attend_and_excite_grad / attend_and_excite     EulerEDMSampler.attend_and_excite          sampling.py:233-252
                                                 (c_noise from get_c_noise :224-231; the network sees the RAW x — the reference calls
                                                  model.model(x, c_noise, cond) without the denoiser's c_in scaling — concatenated with
                                                  cond["concat"], wrappers.py:23-35; only the CONDITIONAL batch; the loss is
                                                  FullLoss.get_min_local_loss over the t_attn maps of size >= min_attn_size,
                                                  loss.py:192-235)
  slice_forward                                  one ResBlock + one SpatialTransformer (openaimodel.py:163-250, attention.py:342-411):
                                                 the sub-stack the first HIP backward slice covers, with the same loss on its t_attn map
                                                 plus a linear functional <G, output> that stands for the gradient arriving from the
                                                 layers downstream

Pinned by tests/test_oracle_golden.py against tests/golden/aae_golden.npz (G13: the gradient torch.autograd.grad returned inside the
REAL reference's attend_and_excite, tests/golden/make_golden.py --g13).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import nets, sampling
from .spec import EngineConfig

SD = Dict[str, torch.Tensor]


def c_noise_of(sd: SD, sigma: torch.Tensor) -> torch.Tensor:
    """EulerEDMSampler.get_c_noise (sampling.py:224-231): quantise sigma to the table, EpsScaling's c_noise = sigma, quantise to its index"""
    table = sd["denoiser.sigmas"]
    sq = table[sampling.sigma_to_idx(table, sigma)]
    return sampling.sigma_to_idx(table, sq)


def local_loss_of(sd: SD, cfg: EngineConfig, x: torch.Tensor, c_noise: torch.Tensor, cond: dict, mask: torch.Tensor,
                  seg_mask: torch.Tensor, min_attn_size: int = 16) -> torch.Tensor:
    maps: list = []
    nets.unet_forward(sd, torch.cat((x, cond["concat"]), dim=1), c_noise, cond["t_crossattn"], cfg.unet, attn_maps=maps)
    return sampling.min_local_loss(maps, mask, seg_mask, sd["loss_fn.g_kernel"], min_attn_size)


def attend_and_excite_grad(sd: SD, cfg: EngineConfig, x: torch.Tensor, sigma: torch.Tensor, cond: dict, mask: torch.Tensor,
                           seg_mask: torch.Tensor, min_attn_size: int = 16) -> Tuple[torch.Tensor, torch.Tensor]:
    """(local_loss [B], d sum(local_loss) / d x [B, 4, h, w]); the samples are independent, so row b of the gradient is the gradient
    of sample b's loss (the reference's implicit-gradient call only accepts B = 1)"""
    c_noise = c_noise_of(sd, sigma)
    with torch.enable_grad():
        xg = x.detach().clone().requires_grad_(True)
        loss = local_loss_of(sd, cfg, xg, c_noise, cond, mask, seg_mask, min_attn_size)
        (g,) = torch.autograd.grad(loss.sum(), [xg])
    return loss.detach(), g


def maps_functional_grad(sd: SD, cfg: EngineConfig, x: torch.Tensor, sigma: torch.Tensor, cond: dict, weights_of, min_attn_size: int = 16):
    """(value, d value / d x) of the smooth functional sum_k <R_k, attn_map_k> / count over the counted t_attn maps, R_k =
    weights_of(shape, k): the same reverse pass as attend-and-excite with a DENSE cotangent on every map (golden G13s)"""
    c_noise = c_noise_of(sd, sigma)
    with torch.enable_grad():
        xg = x.detach().clone().requires_grad_(True)
        maps: list = []
        nets.unet_forward(sd, torch.cat((xg, cond["concat"]), dim=1), c_noise, cond["t_crossattn"], cfg.unet, attn_maps=maps)
        used = [m for m in maps if m["name"].endswith("t_attn") and m["size"] >= min_attn_size]
        val = sum((weights_of(m["attn_map"].shape, k) * m["attn_map"]).sum() for k, m in enumerate(used)) / len(used)
        (g,) = torch.autograd.grad(val, [xg])
    return val.detach(), g


def attend_and_excite(sd: SD, cfg: EngineConfig, x: torch.Tensor, sigma: torch.Tensor, cond: dict, mask: torch.Tensor,
                      seg_mask: torch.Tensor, alpha: float, iter_enabled: bool, thres: float, max_iter: int = 20) -> torch.Tensor:
    """the update loop of sampling.py:240-252: x <- x - alpha * grad, repeated while iter_enabled and loss > thres and iters <= max_iter"""
    iters = 0
    while True:
        loss, g = attend_and_excite_grad(sd, cfg, x, sigma, cond, mask, seg_mask)
        x = x - alpha * g
        iters += 1
        if not iter_enabled or bool((loss <= thres).all()) or iters > max_iter:
            return x


# ------------------------------------------------------------------------------------------------ the two-block slice
def slice_forward(sd: SD, res_prefix: str, st_prefix: str, h0: torch.Tensor, emb: torch.Tensor, ctx: torch.Tensor, heads: int):
    """ResBlock(res_prefix) -> SpatialTransformer(st_prefix) on h0 [B, C, h, w]; returns (output, [t_attn map item])"""
    maps: list = []
    h1 = nets._resblock(sd, res_prefix, h0, emb)
    h2 = nets._spatial_transformer(sd, st_prefix, h1, ctx, heads, "slice.", maps)
    return h2, maps


def slice_loss_and_grad(sd: SD, res_prefix: str, st_prefix: str, h0: torch.Tensor, emb: torch.Tensor, ctx: torch.Tensor, heads: int,
                        mask: torch.Tensor, seg_mask: torch.Tensor, cot: Optional[torch.Tensor], g_kernel: torch.Tensor,
                        min_attn_size: int = 1):
    """loss_b = min_local_loss(t_attn map of the slice)_b + <cot_b, output_b>; returns (local loss [B], output, d sum(loss) / d h0)"""
    with torch.enable_grad():
        hg = h0.detach().clone().requires_grad_(True)
        out, maps = slice_forward(sd, res_prefix, st_prefix, hg, emb, ctx, heads)
        ll = sampling.min_local_loss(maps, mask, seg_mask, g_kernel, min_attn_size)
        total = ll.sum() + ((cot * out).sum() if cot is not None else 0.0)
        (g,) = torch.autograd.grad(total, [hg])
    return ll.detach(), out.detach(), g
