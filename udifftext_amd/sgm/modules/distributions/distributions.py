"""Diagonal Gaussian posterior of the KL autoencoder (reference sgm/modules/distributions/distributions.py:24-41).

``sample()`` keeps the reference's RNG contract: the noise is drawn with ``torch.randn`` on the CPU default
generator (``udifftext_amd.rng.randn``; per-image generators under sharded sampling) and moved to the device; the arithmetic runs in the HIP kernel ``udt_posterior_sample``.
"""
import torch

from udifftext_amd import ops, rng


class DiagonalGaussianDistribution(object):
    def __init__(self, moments_nhwc: torch.Tensor, deterministic: bool = False):
        """moments_nhwc: fp32 [B, h, w, >=8] (mean channels 0..3, logvar channels 4..7)"""
        self.parameters = moments_nhwc
        self.deterministic = deterministic

    @property
    def mean(self):
        return ops.nhwc_to_nchw(self.parameters, 8)[:, :4]

    def sample(self, scale: float = 1.0) -> torch.Tensor:
        B, h, w, _ = self.parameters.shape
        noise = rng.randn_on((B, 4, h, w), self.parameters.device)
        if self.deterministic:
            noise = torch.zeros_like(noise)
        return ops.posterior_sample(self.parameters, noise, scale)

    def mode(self):
        return self.mean
