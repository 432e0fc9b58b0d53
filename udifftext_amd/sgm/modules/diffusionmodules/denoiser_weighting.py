"""Loss weightings; instantiated by configs/test/textdesign_sd_2.yaml:17-18 but used only in training
(reference sgm/modules/diffusionmodules/denoiser_weighting.py:22-24)."""
import torch


class UnitWeighting:
    def __call__(self, sigma):
        return torch.ones_like(sigma)


class EpsWeighting:
    def __call__(self, sigma):
        return sigma ** -2.0
