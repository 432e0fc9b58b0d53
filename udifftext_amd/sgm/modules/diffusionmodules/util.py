"""Small numerics helpers (reference sgm/modules/diffusionmodules/util.py:19-32,206-230,233-239)."""
import math

import torch


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2):
    if schedule != "linear":
        raise NotImplementedError(schedule)
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2
    return betas.numpy()


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    """[cos(t f) | sin(t f)], f_k = exp(-ln(max_period) k / (dim/2)); torch version for host-side use —
    the UNet itself uses the HIP kernel ``udt_timestep_embedding``."""
    if repeat_only:
        return timesteps[:, None].repeat(1, dim)
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half).to(timesteps.device)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module
