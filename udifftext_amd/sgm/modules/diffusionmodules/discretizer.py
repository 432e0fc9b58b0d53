"""Sigma discretisation (host-side, fp64 table -> fp32 sigmas).

Follows reference sgm/modules/diffusionmodules/discretizer.py:10-13,16-20,41-68 and
make_beta_schedule (sgm/modules/diffusionmodules/util.py:19-32): linear-beta DDPM schedule,
``sigma = sqrt((1 - abar) / abar)`` cast to fp32 before the square root, n-step sub-sampling by
``linspace(999, 0, n, endpoint=False).astype(int)[::-1]``.
"""
from __future__ import annotations

import numpy as np
import torch

from ...util import append_zero
from .util import make_beta_schedule


def generate_roughly_equally_spaced_steps(num_substeps: int, max_step: int) -> np.ndarray:
    return np.linspace(max_step - 1, 0, num_substeps, endpoint=False).astype(int)[::-1]


class Discretization:
    def __call__(self, n, do_append_zero=True, device="cpu", flip=False):
        sigmas = self.get_sigmas(n, device=device)
        if do_append_zero:
            sigmas = append_zero(sigmas)
        return torch.flip(sigmas, (0,)) if flip else sigmas

    def get_sigmas(self, n, device):
        raise NotImplementedError


class LegacyDDPMDiscretization(Discretization):
    def __init__(self, linear_start=0.00085, linear_end=0.0120, num_timesteps=1000):
        self.num_timesteps = num_timesteps
        betas = make_beta_schedule("linear", num_timesteps, linear_start=linear_start, linear_end=linear_end)
        self.alphas_cumprod = np.cumprod(1.0 - betas, axis=0)

    def get_sigmas(self, n, device="cpu"):
        if n < self.num_timesteps:
            abar = self.alphas_cumprod[generate_roughly_equally_spaced_steps(n, self.num_timesteps)]
        elif n == self.num_timesteps:
            abar = self.alphas_cumprod
        else:
            raise ValueError(f"cannot sub-sample {n} steps from {self.num_timesteps}")
        ratio = torch.tensor((1 - abar) / abar, dtype=torch.float32, device=device)
        return torch.flip(ratio ** 0.5, (0,))
