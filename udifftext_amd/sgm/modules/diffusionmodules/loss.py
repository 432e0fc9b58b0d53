"""Loss objects instantiated by ``loss_fn_config`` (configs/test/textdesign_sd_2.yaml:109-131).

Only what inference touches is implemented (reference sgm/modules/diffusionmodules/loss.py):
``FullLoss.__init__`` (:73-101, builds the sigma sampler and the ``g_kernel`` buffer — a state-dict entry),
``get_gaussian_kernel`` (:103-129) and ``get_min_local_loss`` (:192-235, scores how well the t_attn maps of the
characters land inside the mask; used by the initial-noise search).  Training losses are out of scope.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from udifftext_amd import ops

from ...util import instantiate_from_config


class StandardDiffusionLoss(nn.Module):
    def __init__(self, sigma_sampler_config, type="l2", offset_noise_level=0.0, batch2model_keys=None):
        super().__init__()
        assert type in ["l2", "l1"]
        self.sigma_sampler = instantiate_from_config(sigma_sampler_config)
        self.type = type
        self.offset_noise_level = offset_noise_level

    def __call__(self, *args, **kwargs):
        raise NotImplementedError("training losses are out of scope of the MI355X inference path")


class FullLoss(StandardDiffusionLoss):
    def __init__(self, seq_len=12, kernel_size=3, gaussian_sigma=0.5, min_attn_size=16, lambda_local_loss=0.0,
                 lambda_ocr_loss=0.0, lambda_style_loss=0.0, ocr_enabled=False, style_enabled=False,
                 predictor_config=None, *args, **kwarg):
        super().__init__(*args, **kwarg)
        if kernel_size != 3:
            raise NotImplementedError("udt_local_loss implements the 3x3 blur of the UDiffText config")
        self.gaussian_kernel_size = kernel_size
        self.register_buffer("g_kernel", self.get_gaussian_kernel(kernel_size, gaussian_sigma, seq_len))
        self.min_attn_size = min_attn_size
        self.lambda_local_loss, self.lambda_ocr_loss, self.lambda_style_loss = lambda_local_loss, lambda_ocr_loss, lambda_style_loss
        self.style_enabled, self.ocr_enabled = style_enabled, ocr_enabled
        if ocr_enabled:
            self.predictor = instantiate_from_config(predictor_config)

    @staticmethod
    def get_gaussian_kernel(kernel_size=3, sigma=1, out_channels=3):
        ax = torch.arange(kernel_size).float() - (kernel_size - 1) / 2.0
        g = torch.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / (2 * sigma ** 2.)) / (2. * torch.pi * sigma ** 2.)
        g = g / g.sum()
        return g.view(1, 1, kernel_size, kernel_size).tile(out_channels, 1, 1, 1)

    def get_min_local_loss(self, attn_map_cache, mask, seg_mask, cond_only: bool = False):
        """-> fp32 [n], n = batch of the attention maps (uncond ‖ cond).  mask [B,1,H,W], seg_mask [B, seg_l] with
        n a multiple of B: sample i of the maps is scored against mask[i % B] (for B = 1 this is the reference's
        broadcast; for B > 1 the reference is undefined — SURVEY.md §8a row N — and this is the per-sample rule).
        cond_only: score only the conditional half (the half every caller keeps, reference sampling.py: ``local_loss[B:]``)
        -> fp32 [n / 2].  One launch per attention map, whatever n is."""
        mask = mask.float().contiguous()
        seg = seg_mask.float().contiguous()
        B = mask.shape[0]
        gk = self.g_kernel[0, 0].reshape(9).float().contiguous()
        loss, count = None, 0
        for item in attn_map_cache:
            if not item["name"].endswith("t_attn") or item["attn_map"] is None:
                continue
            heads, size, am = item["heads"], item["size"], item["attn_map"]
            if size < self.min_attn_size:
                continue
            n = am.shape[0] // heads
            assert n % B == 0 and seg.shape[1] <= am.shape[2]
            first = n // 2 if cond_only else 0
            assert first % B == 0
            if loss is None:
                loss = torch.zeros((n - first,), dtype=torch.float32, device=am.device)
            ops.local_loss_accumulate(am[first * heads:], mask, seg, gk, loss, heads, size)
            count += 1
        return loss / count
