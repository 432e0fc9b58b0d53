"""Functional fp32 restatement of the diffusion plumbing around the UNet (TEST INFRASTRUCTURE).

  ddpm_sigmas / denoiser_sigma_table   LegacyDDPMDiscretization   sgm/modules/diffusionmodules/discretizer.py:10-13,41-68
                                       make_beta_schedule         sgm/modules/diffusionmodules/util.py:19-32
  sigma_to_idx                         DiscreteDenoiser           sgm/modules/diffusionmodules/denoiser.py:49-63
  denoise_cfg                          VanillaCFG + Denoiser.__call__ + EpsScaling
                                       guiders.py:25-40, denoiser.py:22-28, denoiser_scaling.py:16-22, wrappers.py:23-35
  euler_sample                         EulerEDMSampler.__call__ / sampler_step   sampling.py:48-59,324-420
  get_init_noise                       EulerEDMSampler.get_init_noise            sampling.py:264-322
  min_local_loss / gaussian_kernel     FullLoss                                  loss.py:103-129,192-235
  conditioning                         GeneralConditioner.get_unconditional_conditioning + embedders
                                       encoders/modules.py:154-217,843-857,1011-1014
  predict                              test.py:19-40
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import nets
from .spec import EngineConfig

SD = Dict[str, torch.Tensor]


# -------------------------------------------------------------------------------------- discretisation
def _alphas_cumprod(num_timesteps: int = 1000, linear_start: float = 0.00085, linear_end: float = 0.0120) -> np.ndarray:
    betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, num_timesteps, dtype=torch.float64) ** 2).numpy()
    return np.cumprod(1.0 - betas, axis=0)


def ddpm_sigmas(n: int, append_zero: bool = True, flip: bool = False, num_timesteps: int = 1000) -> torch.Tensor:
    """n-step sigma table, descending (then optionally 0 appended / flipped), fp32."""
    ac = _alphas_cumprod(num_timesteps)
    if n < num_timesteps:
        steps = np.linspace(num_timesteps - 1, 0, n, endpoint=False).astype(int)[::-1]
        ac = ac[steps]
    elif n != num_timesteps:
        raise ValueError
    sig = torch.tensor((1 - ac) / ac, dtype=torch.float32) ** 0.5
    sig = torch.flip(sig, (0,))
    if append_zero:
        sig = torch.cat([sig, sig.new_zeros([1])])
    return torch.flip(sig, (0,)) if flip else sig


def denoiser_sigma_table(num_idx: int = 1000) -> torch.Tensor:
    """DiscreteDenoiser.sigmas buffer: ascending, no zero (denoiser.py:43-46)."""
    return ddpm_sigmas(num_idx, append_zero=False, flip=True)


def sigma_to_idx(table: torch.Tensor, sigma: torch.Tensor) -> torch.Tensor:
    return (sigma - table[:, None]).abs().argmin(dim=0).view(sigma.shape)


# ------------------------------------------------------------------------------------------- denoising
def denoise_cfg(sd: SD, cfg: EngineConfig, x: torch.Tensor, sigma: torch.Tensor, cond: dict, uc: dict, scale: float,
                attn_maps: Optional[list] = None) -> torch.Tensor:
    """One guided denoiser evaluation: returns D(x, sigma) after classifier-free guidance."""
    table = sd["denoiser.sigmas"]
    xx = torch.cat([x] * 2)
    ss = torch.cat([sigma] * 2)
    c = {k: torch.cat((uc[k], cond[k]), 0) for k in cond}
    sq = table[sigma_to_idx(table, ss)]                      # possibly_quantize_sigma
    sq4 = sq[:, None, None, None]
    c_in = 1 / (sq4 ** 2 + 1.0) ** 0.5
    c_out = -sq4
    c_noise = sigma_to_idx(table, sq)                        # integer timestep
    net_in = torch.cat((xx * c_in, c["concat"]), dim=1)
    eps = nets.unet_forward(sd, net_in, c_noise, c["t_crossattn"], cfg.unet, attn_maps=attn_maps)
    den = eps * c_out + xx
    x_u, x_c = den.chunk(2)
    return x_u + scale * (x_c - x_u)


def gaussian_kernel(kernel_size: int = 3, sigma: float = 1.0, channels: int = 12) -> torch.Tensor:
    ax = torch.arange(kernel_size).float() - (kernel_size - 1) / 2.0
    g = torch.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / (2 * sigma ** 2)) / (2 * torch.pi * sigma ** 2)
    g = g / g.sum()
    return g.view(1, 1, kernel_size, kernel_size).tile(channels, 1, 1, 1)


def min_local_loss(attn_maps: List[dict], mask: torch.Tensor, seg_mask: torch.Tensor, g_kernel: torch.Tensor,
                   min_attn_size: int = 16) -> torch.Tensor:
    loss, count = 0, 0
    for item in attn_maps:
        if not item["name"].endswith("t_attn") or item["size"] < min_attn_size:
            continue
        heads, size, am = item["heads"], item["size"], item["attn_map"]
        seg_l = seg_mask.shape[1]
        bh, n, l = am.shape
        am = am.reshape(-1, heads, n, l)[..., :seg_l].permute(0, 1, 3, 2).mean(dim=1)       # b, l, n
        am = F.conv2d(am.reshape(-1, seg_l, size, size), g_kernel, padding=1, groups=seg_l).reshape(-1, seg_l, n)
        mm = F.interpolate(mask, (size, size)).tile((1, seg_l, 1, 1)).reshape(-1, seg_l, n)
        p = (mm * am).max(dim=-1)[0] + (1 - seg_mask)
        loss = loss + (-p.min(dim=-1)[0])
        count += 1
    return loss / count


def euler_sample(sd: SD, cfg: EngineConfig, x: torch.Tensor, cond: dict, uc: dict, num_steps: int, scale: float,
                 trajectory: Optional[list] = None) -> torch.Tensor:
    """Deterministic Euler (== DDIM eta 0) loop with CFG; x is the unit-variance initial noise."""
    sigmas = ddpm_sigmas(num_steps)
    x = x * torch.sqrt(1.0 + sigmas[0] ** 2.0)
    s_in = x.new_ones([x.shape[0]])
    for i in range(num_steps):
        sigma, nxt = s_in * sigmas[i], s_in * sigmas[i + 1]
        den = denoise_cfg(sd, cfg, x, sigma, cond, uc, scale)
        d = (x - den) / sigma[:, None, None, None]
        x = x + d * (nxt - sigma)[:, None, None, None]
        if trajectory is not None:
            trajectory.append(x.clone())
    return x


def get_init_noise(sd: SD, cfg: EngineConfig, shape: Tuple[int, ...], cond: dict, uc: dict, batch: dict, noise_iters: int,
                   scale: float) -> Tuple[torch.Tensor, list]:
    """noise_iters candidate noises, each scored by the local loss after the 2nd of 2 Euler steps; returns
    the arg-min (or the single draw when noise_iters == 0).  Draw order: noise_iters+1 CPU randn calls."""
    randn = torch.randn(shape)
    scored = []
    g_kernel = sd["loss_fn.g_kernel"]
    for _ in range(noise_iters):
        sigmas = ddpm_sigmas(2)
        x = randn.clone() * torch.sqrt(1.0 + sigmas[0] ** 2.0)
        s_in = x.new_ones([x.shape[0]])
        last = None
        for i in range(2):
            sigma, nxt = s_in * sigmas[i], s_in * sigmas[i + 1]
            maps: list = []
            den = denoise_cfg(sd, cfg, x, sigma, cond, uc, scale, attn_maps=maps)
            ll = min_local_loss(maps, batch["mask"], batch["seg_mask"], g_kernel)
            last = ll[ll.shape[0] // 2:]
            d = (x - den) / sigma[:, None, None, None]
            x = x + d * (nxt - sigma)[:, None, None, None]
        scored.append((randn, last.item()))
        randn = torch.randn(shape)
    scored.sort(key=lambda t: t[-1])
    if scored:
        return scored[0][0], [s for _, s in scored]
    return randn, []


# ----------------------------------------------------------------------------------------- conditioning
def conditioning(sd: SD, cfg: EngineConfig, batch: dict) -> Tuple[dict, dict]:
    """(c, uc) with uc's label embedding zeroed.  Two CPU randn draws of shape [B,4,h,w]: c first, then uc."""
    out = []
    for which in ("c", "uc"):
        labels = batch["label"] if which == "c" else ["" for _ in batch["label"]]
        t = nets.label_encoder(sd, labels, cfg.label)
        if which == "uc":
            t = torch.zeros_like(t)
        m = F.interpolate(batch["mask"], scale_factor=0.125, mode="bilinear")
        mom = nets.vae_encode_moments(sd, batch["masked"], cfg.vae, prefix="conditioner.embedders.2.model.")
        z = cfg.scale_factor * nets.posterior_sample(mom, torch.randn(mom.shape[0], 4, *mom.shape[2:]))
        out.append({"t_crossattn": t, "concat": torch.cat((m, z), dim=1)})
    return out[0], out[1]


def predict(sd: SD, cfg: EngineConfig, batch: dict, steps: int, scale: float = 5.0, noise_iters: int = 0):
    """test.py:19-40 on a CPU batch dict; returns (samples in [0,1], latent)."""
    with torch.no_grad():
        c, uc = conditioning(sd, cfg, batch)
        H, W = (int(v) for v in batch["target_size_as_tuple"][0])
        B = batch["image"].shape[0]
        x, _ = get_init_noise(sd, cfg, (B, 4, H // 8, W // 8), c, uc, batch, noise_iters, scale)
        z = euler_sample(sd, cfg, x, c, uc, steps, scale)
        img = nets.vae_decode(sd, z / cfg.scale_factor, cfg.vae)
        return torch.clamp((img + 1.0) / 2.0, 0.0, 1.0), z
