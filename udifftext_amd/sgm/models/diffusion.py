"""``DiffusionEngine``: the object the reference's ``init_model`` builds from ``configs/*.yaml`` and that the
sampler reaches into (``model.denoiser``, ``model.model``, ``model.conditioner``, ``model.first_stage_model``,
``model.loss_fn``) — reference sgm/models/diffusion.py:22-136.  A plain ``nn.Module`` (no Lightning).  Round 6: the training half's
core — ``forward`` (the loss), ``shared_step`` / ``training_step`` and ``configure_optimizers`` (:138-172,202-222) — runs on the
HIP path's written-out reverse pass (udifftext_amd.training); EMA, logging and log_images stay out of scope.
"""
from __future__ import annotations

from typing import List, Union

import torch
import torch.nn as nn

from ..modules import UNCONDITIONAL_CONFIG
from ..modules.diffusionmodules.wrappers import OPENAIUNETWRAPPER
from ..util import default, disabled_train, get_obj_from_str, instantiate_from_config
from .autoencoder import _load_checkpoint


class DiffusionEngine(nn.Module):
    def __init__(self, network_config, denoiser_config, first_stage_config, conditioner_config=None, sampler_config=None,
                 optimizer_config=None, scheduler_config=None, loss_fn_config=None, network_wrapper=None,
                 ckpt_path: Union[None, str] = None, use_ema: bool = False, ema_decay_rate: float = 0.9999,
                 scale_factor: float = 1.0, disable_first_stage_autocast=False, input_key: str = "jpg",
                 log_keys: Union[List, None] = None, no_cond_log: bool = False, compile_model: bool = False,
                 opt_keys: Union[List, None] = None):
        super().__init__()
        if use_ema:
            raise NotImplementedError("EMA weights are a training feature (out of scope)")
        self.opt_keys, self.log_keys, self.input_key = opt_keys, log_keys, input_key
        self.optimizer_config = default(optimizer_config, {"target": "torch.optim.AdamW"})
        network = instantiate_from_config(network_config)
        self.model = get_obj_from_str(default(network_wrapper, OPENAIUNETWRAPPER))(network, compile_model=compile_model)
        self.denoiser = instantiate_from_config(denoiser_config)
        self.sampler = instantiate_from_config(sampler_config) if sampler_config is not None else None
        self.conditioner = instantiate_from_config(default(conditioner_config, UNCONDITIONAL_CONFIG))
        self.scheduler_config = scheduler_config
        self._init_first_stage(first_stage_config)
        self.loss_fn = instantiate_from_config(loss_fn_config) if loss_fn_config is not None else None
        self.use_ema = False
        self.scale_factor = scale_factor
        self.disable_first_stage_autocast = disable_first_stage_autocast
        self.no_cond_log = no_cond_log
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path)

    def init_from_ckpt(self, path: str):
        """reference diffusion.py:87-105 (.ckpt / .safetensors, strict=False, prints the key report); also returns
        (missing, unexpected)"""
        sd = _load_checkpoint(path)
        missing, unexpected = self.load_state_dict(sd, strict=False)
        print(f"Restored from {path} with {len(missing)} missing and {len(unexpected)} unexpected keys")
        if missing:
            print(f"Missing Keys: {missing}")
        if unexpected:
            print(f"Unexpected Keys: {unexpected}")
        return missing, unexpected

    def _check_vae_alias(self, sd) -> None:
        """prepare(dedup_vae=True) pointed the LatentEncoder at first_stage_model because their weights were equal; a later
        checkpoint whose two prefixes DIFFER needs two modules again: split them before loading (and say so)"""
        alias = getattr(self, "_vae_alias_prefixes", None)
        if not alias:
            return
        a, b = alias
        differ = any(k.startswith(a) and (b + k[len(a):]) in sd and not torch.equal(sd[k].cpu(), sd[b + k[len(a):]].cpu()) for k in sd)
        # a checkpoint with only ONE of the two prefixes (the reference's training flow loads first_stage_model.* alone,
        # diffusion.py:87-105) must not be written through the alias into the other module
        has_a = any(k.startswith(a) for k in sd)
        has_b = any(k.startswith(b) for k in sd)
        if differ or (has_a != has_b):
            import copy
            import warnings
            warnings.warn(("the checkpoint holds DIFFERENT weights under %s and %s" % (a, b) if differ else
                           "the checkpoint holds only one of %s / %s" % (a, b)) + ": undoing the VAE dedup of prepare()")
            for emb in self.conditioner.embedders:
                if getattr(emb, "model", None) is self.first_stage_model:
                    emb.model = copy.deepcopy(self.first_stage_model)
            self._vae_alias_prefixes = None

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self._check_vae_alias(state_dict)
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def prepare(self, free_masters: bool = False, dedup_vae: bool = True):
        """load-time packing of the device weight layouts, VAE dedup, optional release of the fp32 masters
        (udifftext_amd.prepare)"""
        from udifftext_amd.prepare import prepare
        return prepare(self, free_masters=free_masters, dedup_vae=dedup_vae)

    def freeze(self):
        for p in self.parameters():
            p.requires_grad_(False)

    def _init_first_stage(self, config):
        model = instantiate_from_config(config).eval()
        model.train = disabled_train
        for p in model.parameters():
            p.requires_grad = False
        self.first_stage_model = model

    def get_input(self, batch):
        return batch[self.input_key]

    @torch.no_grad()
    def decode_first_stage(self, z):
        return self.first_stage_model.decode(1.0 / self.scale_factor * z)

    @torch.no_grad()
    def encode_first_stage(self, x):
        return self.scale_factor * self.first_stage_model.encode(x)

    # ------------------------------------------------------------------------------ training (reference :138-172,202-222)
    def forward(self, x, batch):
        """reference :138-142: ``loss_fn(model, denoiser, conditioner, x, batch, ...)`` -> (loss, loss_dict).  x: latents
        [B, 4, h, w]; batch: ``label`` / ``mask`` / ``masked`` for the conditioner (its ucg draw included) plus ``seg`` / ``seg_mask``"""
        from udifftext_amd import training
        cond = self.conditioner(batch)
        loss_dict, _ = training.training_loss_and_grads(self, x, cond, batch["seg"], batch["seg_mask"], want_grads=False)
        return loss_dict["loss/full_loss"], loss_dict

    def shared_step(self, batch):
        """reference :144-149 (latents from the first stage, then the loss) — with the gradients of the trained parameters"""
        from udifftext_amd import training
        x = self.encode_first_stage(self.get_input(batch))
        cond = self.conditioner(batch)
        return training.training_loss_and_grads(self, x, cond, batch["seg"], batch["seg_mask"])

    def configure_optimizers(self, learning_rate: float = 5.0e-5):
        """reference :202-222: AdamW (the default optimizer_config) over the parameters whose names contain an ``opt_keys`` entry;
        the LambdaLR 0.95^epoch is ``optimizer.set_epoch``"""
        from udifftext_amd import training
        if self.optimizer_config.get("target", "torch.optim.AdamW") != "torch.optim.AdamW":
            raise NotImplementedError("udt_adamw_f32 implements the reference's default optimiser (torch.optim.AdamW)")
        named = training.trainable_parameters(self)
        if not named:
            raise ValueError("opt_keys selects no parameter (configs/train/textdesign_sd_2.yaml: t_attn, t_norm)")
        return training.AdamW(named, lr=learning_rate, **self.optimizer_config.get("params", {}))

    def training_step(self, batch, optimizer, dist=None):
        """reference :151-172 + the optimiser step Lightning takes after it: loss, gradients, rank average (``dist``), AdamW update;
        returns the loss dict"""
        from udifftext_amd import training
        loss_dict, grads = self.shared_step(batch)
        training.allreduce_gradients(grads, [n for n, _ in optimizer.named], dist)
        optimizer.step(grads)
        return loss_dict
