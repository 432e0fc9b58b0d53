"""Preconditioning coefficients (reference sgm/modules/diffusionmodules/denoiser_scaling.py:16-22)."""
import torch


class EpsScaling:
    """eps-prediction: c_skip = 1, c_out = -sigma, c_in = (sigma^2 + 1)^-1/2, c_noise = sigma"""

    def __call__(self, sigma: torch.Tensor):
        return torch.ones_like(sigma), -sigma, 1 / (sigma ** 2 + 1.0) ** 0.5, sigma.clone()


class VScaling:
    def __call__(self, sigma: torch.Tensor):
        denom = sigma ** 2 + 1.0
        return 1.0 / denom, -sigma / denom ** 0.5, 1.0 / denom ** 0.5, sigma.clone()
