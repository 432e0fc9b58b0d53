"""Guidance combine and the Euler direction (reference sgm/modules/diffusionmodules/sampling_utils.py:7-9,39-40)."""
from ...util import append_dims


class NoDynamicThresholding:
    def __call__(self, uncond, cond, scale):
        return uncond + scale * (cond - uncond)


def to_d(x, sigma, denoised):
    return (x - denoised) / append_dims(sigma, x.ndim)
