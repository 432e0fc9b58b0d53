"""The training loss of the text cross-attention on the CPU oracle, with torch.autograd parameter gradients (second half;
TEST INFRASTRUCTURE — only tests/ import this).

  local_loss            FullLoss.get_local_loss                      sgm/modules/diffusionmodules/loss.py:237-286
  training_loss         FullLoss.__call__ (ocr / style off)          loss.py:131-176  (get_diff_loss :60-71; DiscreteDenoiser.__call__
                                                                     denoiser.py:22-28 with EpsScaling / EpsWeighting)
  training_grads        torch.autograd.grad of loss/full_loss with respect to the parameters DiffusionEngine.configure_optimizers
                        selects (sgm/models/diffusion.py:202-217: names containing an opt_keys entry — t_attn, t_norm)
  maps_functional_param_grads   the same reverse pass from a DENSE cotangent on the counted t_attn maps (golden G14s)

Pinned by tests/test_oracle_golden.py against tests/golden/train_golden.npz (the REAL reference's FullLoss.__call__ under autograd,
tests/golden/make_golden.py --g14).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import nets
from .spec import EngineConfig

SD = Dict[str, torch.Tensor]
OPT_KEYS = ("t_attn", "t_norm")


def local_loss(attn_maps: List[dict], seg: torch.Tensor, seg_mask: torch.Tensor, g_kernel: torch.Tensor, min_attn_size: int = 16) -> torch.Tensor:
    loss, count = 0, 0
    for item in attn_maps:
        if not item["name"].endswith("t_attn") or item["size"] < min_attn_size:
            continue
        heads, size, am = item["heads"], item["size"], item["attn_map"]
        seg_l = seg_mask.shape[1]
        bh, n, l = am.shape
        am = am.reshape(-1, heads, n, l)[..., :seg_l].permute(0, 1, 3, 2).mean(dim=1)
        am = F.conv2d(am.reshape(-1, seg_l, size, size), g_kernel, padding=1, groups=seg_l).reshape(-1, seg_l, n)
        sm = F.interpolate(seg, (size, size)).reshape(-1, seg_l, n)
        p = (sm * am).max(dim=-1)[0] * seg_mask
        nn_ = ((1 - sm) * am).max(dim=-1)[0] * seg_mask
        loss = loss + (nn_.sum(dim=-1) / seg_mask.sum(dim=-1) - p.sum(dim=-1) / seg_mask.sum(dim=-1))
        count += 1
    return loss / count


def trainable_names(sd: SD) -> List[str]:
    return [k for k in sd if k.startswith("model.") and any(key in k for key in OPT_KEYS)]


def _forward(sd: SD, cfg: EngineConfig, z, cond, sigma_idx, noise, maps):
    table = sd["denoiser.sigmas"]
    sigma = table[sigma_idx][:, None, None, None]
    noised = z + noise * sigma
    c_in = 1 / (sigma ** 2 + 1.0) ** 0.5
    eps = nets.unet_forward(sd, torch.cat((noised * c_in, cond["concat"]), dim=1), sigma_idx, cond["t_crossattn"], cfg.unet, attn_maps=maps)
    return eps * (-sigma) + noised, sigma


def training_loss(sd: SD, cfg: EngineConfig, z, cond, seg, seg_mask, sigma_idx, noise, lambda_local: float = 0.01, min_attn_size: int = 16):
    maps: list = []
    out, sigma = _forward(sd, cfg, z, cond, sigma_idx, noise, maps)
    w = sigma ** -2.0
    diff = torch.mean((w * (out - z) ** 2).reshape(z.shape[0], -1), 1).mean()
    loc = local_loss(maps, seg, seg_mask, sd["loss_fn.g_kernel"], min_attn_size).mean()
    return {"loss/diff_loss": diff, "loss/local_loss": loc, "loss/full_loss": diff + lambda_local * loc}


def training_grads(sd: SD, cfg: EngineConfig, z, cond, seg, seg_mask, sigma_idx, noise, lambda_local: float = 0.01):
    names = trainable_names(sd)
    with torch.enable_grad():
        sdg = dict(sd)
        for n in names:
            sdg[n] = sd[n].detach().clone().requires_grad_(True)
        ld = training_loss(sdg, cfg, z, cond, seg, seg_mask, sigma_idx, noise, lambda_local)
        gs = torch.autograd.grad(ld["loss/full_loss"], [sdg[n] for n in names])
    return {k: v.detach() for k, v in ld.items()}, dict(zip(names, gs))


def maps_functional_param_grads(sd: SD, cfg: EngineConfig, z, cond, sigma_idx, noise, weights_of, min_attn_size: int = 16):
    """value and parameter gradients of sum_k <R_k, attn_map_k> / count over the counted maps of the TRAINING forward (G14s)"""
    names = trainable_names(sd)
    with torch.enable_grad():
        sdg = dict(sd)
        for n in names:
            sdg[n] = sd[n].detach().clone().requires_grad_(True)
        maps: list = []
        _forward(sdg, cfg, z, cond, sigma_idx, noise, maps)
        used = [m for m in maps if m["name"].endswith("t_attn") and m["size"] >= min_attn_size]
        val = sum((weights_of(m["attn_map"].shape, k) * m["attn_map"]).sum() for k, m in enumerate(used)) / len(used)
        gs = torch.autograd.grad(val, [sdg[n] for n in names], allow_unused=True)
    return val.detach(), {n: (g if g is not None else torch.zeros_like(sd[n])) for n, g in zip(names, gs)}
