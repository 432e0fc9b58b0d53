"""``DiffusionEngine``: the object the reference's ``init_model`` builds from ``configs/*.yaml`` and that the
sampler reaches into (``model.denoiser``, ``model.model``, ``model.conditioner``, ``model.first_stage_model``,
``model.loss_fn``) — reference sgm/models/diffusion.py:22-136.  A plain ``nn.Module`` (no Lightning); the
training half of the reference class (training_step / optimisers / EMA / log_images, :138-328) is out of scope.
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

    def forward(self, x, batch):
        raise NotImplementedError("training forward (loss) is out of scope of the MI355X inference path")
