"""Model / runtime configuration helpers.

``default_model_config()`` builds, in code, the same ``target:``/``params:`` tree as the reference's
``configs/test/textdesign_sd_2.yaml`` (minus the ``ckpt_path`` entries — there are no checkpoints in this
environment); ``load_config(path)`` reads any reference YAML (model or runtime) into attribute-accessible
dicts, so the reference's own config files can be used unchanged (it is a minimal stand-in for OmegaConf).
"""
from __future__ import annotations

import copy

import yaml


class AttrDict(dict):
    """dict with attribute access (the subset of OmegaConf's DictConfig the reference's util.py / test.py use)"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(o):
    if isinstance(o, dict):
        return AttrDict({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, list):
        return [_wrap(v) for v in o]
    return o


def load_config(path: str) -> AttrDict:
    with open(path) as f:
        return _wrap(yaml.safe_load(f))


_DDCONFIG = dict(attn_type="vanilla-xformers", double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                 ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
_DISC = {"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"}


def _vae_config() -> dict:
    return {"target": "sgm.models.autoencoder.AutoencoderKLInferenceWrapper",
            "params": {"embed_dim": 4, "monitor": "val/rec_loss", "ddconfig": copy.deepcopy(_DDCONFIG),
                       "lossconfig": {"target": "torch.nn.Identity"}}}


def default_model_config() -> AttrDict:
    model = {
        "target": "sgm.models.diffusion.DiffusionEngine",
        "params": {
            "opt_keys": ["t_attn"], "input_key": "image", "scale_factor": 0.18215, "disable_first_stage_autocast": True,
            "denoiser_config": {
                "target": "sgm.modules.diffusionmodules.denoiser.DiscreteDenoiser",
                "params": {"num_idx": 1000,
                           "weighting_config": {"target": "sgm.modules.diffusionmodules.denoiser_weighting.EpsWeighting"},
                           "scaling_config": {"target": "sgm.modules.diffusionmodules.denoiser_scaling.EpsScaling"},
                           "discretization_config": dict(_DISC)}},
            "network_config": {
                "target": "sgm.modules.diffusionmodules.openaimodel.UnifiedUNetModel",
                "params": {"in_channels": 9, "out_channels": 4, "ctrl_channels": 0, "model_channels": 320,
                           "attention_resolutions": [4, 2, 1], "save_attn_type": ["t_attn"],
                           "save_attn_layers": ["output_blocks.6.1"], "num_res_blocks": 2, "channel_mult": [1, 2, 4, 4],
                           "num_head_channels": 64, "use_linear_in_transformer": True, "transformer_depth": 1,
                           "t_context_dim": 2048}},
            "conditioner_config": {
                "target": "sgm.modules.GeneralConditioner",
                "params": {"emb_models": [
                    {"is_trainable": False, "emb_key": "t_crossattn", "ucg_rate": 0.1, "input_key": "label",
                     "target": "sgm.modules.encoders.modules.LabelEncoder",
                     "params": {"max_len": 12, "emb_dim": 2048, "n_heads": 8, "n_trans_layers": 12}},
                    {"is_trainable": False, "input_key": "mask", "target": "sgm.modules.encoders.modules.SpatialRescaler",
                     "params": {"in_channels": 1, "multiplier": 0.125}},
                    {"is_trainable": False, "input_key": "masked", "target": "sgm.modules.encoders.modules.LatentEncoder",
                     "params": {"scale_factor": 0.18215, "config": _vae_config()}}]}},
            "first_stage_config": _vae_config(),
            "loss_fn_config": {
                "target": "sgm.modules.diffusionmodules.loss.FullLoss",
                "params": {"seq_len": 12, "kernel_size": 3, "gaussian_sigma": 1.0, "min_attn_size": 16,
                           "lambda_local_loss": 0.01, "lambda_ocr_loss": 0.001, "ocr_enabled": False,
                           "predictor_config": {"target": "sgm.modules.predictors.model.ParseqPredictor",
                                                "params": {"ckpt_path": "./checkpoints/predictors/parseq-bb5792a6.pt"}},
                           "sigma_sampler_config": {
                               "target": "sgm.modules.diffusionmodules.sigma_sampling.DiscreteSampling",
                               "params": {"num_idx": 1000, "discretization_config": dict(_DISC)}}}},
        },
    }
    return _wrap({"model": model})


def default_runtime_config(**over) -> AttrDict:
    """the knobs of the reference's configs/test.yaml that the inference path reads"""
    cfg = dict(type="test", channel=4, factor=8, scale=[5.0, 0.0], noise_iters=0, force_uc_zero_embeddings=["label"],
               aae_enabled=False, detailed=False, steps=50, init_step=0, batch_size=1, gpu=0)
    cfg.update(over)
    return _wrap(cfg)
