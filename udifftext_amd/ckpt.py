"""Checkpoint ingestion helpers.

``DiffusionEngine.init_from_ckpt`` / ``load_state_dict(strict=False)`` take UDiffText checkpoints as they are (same 1330
keys as the reference).  The reference's TRAINING starts from Stability's SD-2 inpainting weights instead
(configs/train.yaml:5 ``load_ckpt_path: ./checkpoints/pretrained/512-inpainting-ema.ckpt``, loaded by util.init_model with
``strict=False``, reference sgm/models/diffusion.py:87-105): that file uses the LDM key names, of which

    model.diffusion_model.*   load straight into UnifiedUNetModel — except ``attn2`` / ``norm2`` of every transformer block
                              (text cross-attention over CLIP tokens), which UDiffText replaces by ``t_attn`` / ``t_norm``
                              over the character-level LabelEncoder tokens (attention.py:265-341): dropped, t_attn/t_norm stay
                              at their initialisation (reported as missing, as the reference's load does);
    first_stage_model.*       load straight into the AutoencoderKL ONLY, like the reference's ``load_state_dict(strict=False)``
                              (diffusion.py:87-105): the engine's LatentEncoder holds its own copy of the autoencoder, filled
                              from its own ``ckpt_path`` (AE_inpainting_2.safetensors, configs/*/textdesign_sd_2.yaml:71,91), so
                              ``conditioner.embedders.<i>.model.*`` are reported as missing.  ``mirror_to_latent_encoder=True``
                              (opt-in, NOT reference behaviour) copies the autoencoder onto that twin as well;
    cond_stage_model.* / model_ema.* / the DDPM schedule buffers (betas, alphas_cumprod, ...)   have no counterpart: dropped.

``map_sd2_inpainting`` performs that mapping on a state dict and reports what it did; nothing here touches the GPU.
"""
from __future__ import annotations

import re
from typing import Dict, Iterable, Tuple

import torch

_DROP_PREFIXES = ("cond_stage_model.", "model_ema.")
_SCHEDULE = re.compile(r"^(betas|alphas_cumprod|alphas_cumprod_prev|sqrt_.*|log_one_minus_alphas_cumprod|posterior_.*|"
                       r"logvar|lvlb_weights|scale_factor)$")
_TEXT_XATTN = re.compile(r"\.transformer_blocks\.\d+\.(attn2|norm2)\.")


def map_sd2_inpainting(sd: Dict[str, torch.Tensor], engine_keys: Iterable[str],
                       mirror_to_latent_encoder: bool = False) -> Tuple[Dict[str, torch.Tensor], Dict[str, list]]:
    """LDM-named SD-2 inpainting state dict -> (state dict in UDiffText names, report).
    report = {"loaded", "duplicated_to_latent_encoder", "dropped_text_cross_attention", "dropped_other", "missing"}"""
    engine_keys = list(engine_keys)
    have = set(engine_keys)
    twin = sorted({m.group(0) for k in engine_keys for m in [re.match(r"conditioner\.embedders\.\d+\.model\.", k)] if m})
    out: Dict[str, torch.Tensor] = {}
    rep = {"loaded": [], "duplicated_to_latent_encoder": [], "dropped_text_cross_attention": [], "dropped_other": [], "missing": []}
    for k, v in sd.items():
        if k.startswith(_DROP_PREFIXES) or _SCHEDULE.match(k):
            rep["dropped_other"].append(k)
        elif k.startswith("model.diffusion_model.") and _TEXT_XATTN.search(k):
            rep["dropped_text_cross_attention"].append(k)
        elif k in have:
            out[k] = v
            rep["loaded"].append(k)
            if mirror_to_latent_encoder and k.startswith("first_stage_model."):
                for pre in twin:
                    k2 = pre + k[len("first_stage_model."):]
                    if k2 in have:
                        out[k2] = v
                        rep["duplicated_to_latent_encoder"].append(k2)
        else:
            rep["dropped_other"].append(k)
    rep["missing"] = [k for k in engine_keys if k not in out]
    return out, rep


def load_sd2_inpainting(engine, sd: Dict[str, torch.Tensor], mirror_to_latent_encoder: bool = False) -> Dict[str, list]:
    """map + ``load_state_dict(strict=False)``; shape mismatches raise like any load.  An engine whose LatentEncoder was
    pointed at ``first_stage_model`` by ``prepare(dedup_vae=True)`` gets its own autoencoder back first
    (DiffusionEngine._check_vae_alias), so the conditioning VAE keeps its weights as in the reference."""
    mapped, rep = map_sd2_inpainting(sd, engine.state_dict().keys(), mirror_to_latent_encoder)
    engine.load_state_dict(mapped, strict=False)
    return rep
