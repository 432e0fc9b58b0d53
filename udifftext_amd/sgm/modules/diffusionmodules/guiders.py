"""Classifier-free guidance (reference sgm/modules/diffusionmodules/guiders.py:8-53): the UNet runs on
``uncond ‖ cond`` (uncond first) and the halves are recombined as ``x_u + s (x_c - x_u)``."""
import torch

from ...util import default, instantiate_from_config

_CAT_KEYS = ("vector", "t_crossattn", "v_crossattn", "concat")


class VanillaCFG:
    def __init__(self, scale, dyn_thresh_config=None):
        self.scale = scale
        self.scale_schedule = lambda sigma: self.scale           # step independent
        self.dyn_thresh = instantiate_from_config(default(
            dyn_thresh_config, {"target": "sgm.modules.diffusionmodules.sampling_utils.NoDynamicThresholding"}))

    def __call__(self, x, sigma):
        x_u, x_c = x.chunk(2)
        return self.dyn_thresh(x_u, x_c, self.scale_schedule(sigma))

    def prepare_inputs(self, x, s, c, uc):
        c_out = {}
        for k in c:
            if k in _CAT_KEYS:
                c_out[k] = torch.cat((uc[k], c[k]), 0)
            else:
                assert c[k] == uc[k]
                c_out[k] = c[k]
        return torch.cat([x] * 2), torch.cat([s] * 2), c_out


class IdentityGuider:
    def __call__(self, x, sigma):
        return x

    def prepare_inputs(self, x, s, c, uc):
        return x, s, dict(c)
