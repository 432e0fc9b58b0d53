"""Training-time sigma sampler; constructed by FullLoss (reference loss.py:23, sigma_sampling.py:16-31)."""
import torch

from ...util import default, instantiate_from_config


class DiscreteSampling:
    def __init__(self, discretization_config, num_idx, do_append_zero=False, flip=True):
        self.num_idx = num_idx
        self.sigmas = instantiate_from_config(discretization_config)(num_idx, do_append_zero=do_append_zero, flip=flip)

    def idx_to_sigma(self, idx):
        return self.sigmas[idx]

    def __call__(self, n_samples, rand=None):
        return self.idx_to_sigma(default(rand, torch.randint(0, self.num_idx, (n_samples,))))
