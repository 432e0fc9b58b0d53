"""Drop-in guards and checkpoint ingestion (§8f rank 3), all on CPU:

* the new ``sgm`` instantiates from the REFERENCE's own ``configs/test/textdesign_sd_2.yaml`` and yields the
  reference's 1330 state-dict keys; the reference's ``util.py`` and ``test.py:predict`` import and run unchanged
  under tiny ``omegaconf`` / ``pytorch_lightning`` / ``torchvision`` shims against it (skipped where
  /root/reference does not exist — the GPU box);
* a full-key-set ``.safetensors`` round trip; load-time ``prepare()``: packing, twin-VAE dedup, master release.
"""
import copy
import importlib
import json
import os
import sys
import types
import unittest.mock as mock

import pytest
import torch

import udifftext_amd  # noqa: F401  (puts the sgm mirror on sys.path)
from udifftext_amd import config as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "configs/test/textdesign_sd_2.yaml")),
                               reason="reference tree not present (GPU box)")


def _strip_ckpt(cfg):
    if isinstance(cfg, dict):
        cfg.pop("ckpt_path", None)
        for v in cfg.values():
            _strip_ckpt(v)
    elif isinstance(cfg, list):
        for v in cfg:
            _strip_ckpt(v)


def small_model_config():
    """the reference architecture (same module tree, same 1330 state-dict KEYS) at reduced width, so that a full
    checkpoint is a few MB: UNet 64 channels, LabelEncoder 128-d, VAE 32 channels"""
    cfg = copy.deepcopy(C.default_model_config())
    p = cfg.model.params
    p.network_config.params.update(model_channels=64, t_context_dim=128)
    le = p.conditioner_config.params.emb_models[0].params
    le.update(emb_dim=128)
    for vae in (p.conditioner_config.params.emb_models[2].params.config, p.first_stage_config):
        vae.params.ddconfig.update(ch=32)
    return cfg


@pytest.fixture(scope="module")
def small_engine():
    from sgm.util import instantiate_from_config
    from udifftext_amd import synth
    torch.manual_seed(0)
    eng = instantiate_from_config(small_model_config().model)
    synth.fill_module_(eng)
    return eng


def test_full_key_set_safetensors_round_trip(small_engine, tmp_path):
    """every one of the reference's 1330 state-dict keys written to a .safetensors and loaded back with zero
    missing / unexpected keys and identical tensors"""
    from safetensors.torch import save_file
    from sgm.util import instantiate_from_config
    ref_keys = list(json.load(open(os.path.join(GOLD, "state_dict_keys.json"))).keys())
    sd = small_engine.state_dict()
    assert list(sd.keys()) == ref_keys and len(ref_keys) == 1330
    path = str(tmp_path / "full.safetensors")
    save_file({k: v.contiguous().clone() for k, v in sd.items()}, path)
    fresh = instantiate_from_config(small_model_config().model)
    missing, unexpected = fresh.init_from_ckpt(path)
    assert missing == [] and unexpected == []
    sd2 = fresh.state_dict()
    assert all(torch.equal(sd[k], sd2[k]) for k in ref_keys)
    # a torch .ckpt ({"state_dict": ...}) takes the same path
    ck = str(tmp_path / "full.ckpt")
    torch.save({"state_dict": sd}, ck)
    assert fresh.init_from_ckpt(ck) == ([], [])


def test_prepare_packs_dedups_and_frees(small_engine):
    from sgm.modules import hipnn as H
    from sgm.util import instantiate_from_config
    eng = instantiate_from_config(small_model_config().model)
    eng.load_state_dict(small_engine.state_dict())
    n_keys = len(eng.state_dict())
    # different VAE copies: nothing to deduplicate
    rep = eng.prepare()
    assert rep["vae_deduplicated"] == 0 and rep["packed_modules"] > 300 and rep["packed_bytes"] > 0
    le = eng.conditioner.embedders[2]
    assert le.model is not eng.first_stage_model
    # the reference loads both from the same file: equal tensors -> one copy
    le.model.load_state_dict(eng.first_stage_model.state_dict())
    rep = eng.prepare()
    assert rep["vae_deduplicated"] == 1 and le.model is eng.first_stage_model
    assert len(eng.state_dict()) == n_keys                      # both prefixes still present
    # a checkpoint that carries first_stage_model.* ONLY (the reference's training flow, diffusion.py:87-105) must not reach the
    # LatentEncoder through the alias: the modules are split again and the conditioning VAE keeps its weights
    import warnings
    keep = {k: v.clone() for k, v in le.model.state_dict().items()}
    only_fs = {k: v + 1.0 for k, v in eng.state_dict().items() if k.startswith("first_stage_model.")}
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        eng.load_state_dict(only_fs, strict=False)
    assert any("undoing the VAE dedup" in str(x.message) for x in w)
    assert le.model is not eng.first_stage_model
    assert all(torch.equal(le.model.state_dict()[k], keep[k]) for k in keep)
    k0 = next(iter(keep))
    assert torch.equal(eng.first_stage_model.state_dict()[k0], keep[k0] + 1.0)
    le.model.load_state_dict(eng.first_stage_model.state_dict())
    rep = eng.prepare()
    assert rep["vae_deduplicated"] == 1 and le.model is eng.first_stage_model
    # packed layouts exist for every evaluated module; fused children were not packed on their own
    blk = eng.model.diffusion_model.input_blocks[1][1].transformer_blocks[0]
    # the LayerNorm-folded layouts serve attn1 / GEGLU (UDT_LN_GEMM, default on): prepare() packs THOSE, not the plain ones
    assert H.LN_GEMM and getattr(blk.attn1, "_pkln", None) is not None and getattr(blk.attn1, "_pk", None) is None
    assert getattr(blk.attn1.to_q, "_pk", None) is None
    wf, c_ln, s_ln = blk.attn1.packed_ln(blk.norm1)
    ref_f = (torch.cat([blk.attn1.to_q.weight, blk.attn1.to_k.weight, blk.attn1.to_v.weight]) * blk.norm1.weight[None, :]).to(torch.bfloat16)
    assert torch.equal(wf, ref_f) and c_ln.shape[0] == wf.shape[0] == s_ln.shape[0]
    wqk, wv = blk.attn1.packed()                  # (on demand) one fused q|k|v matrix: the flash kernel reads V row-major
    ref_qk = torch.cat([blk.attn1.to_q.weight, blk.attn1.to_k.weight, blk.attn1.to_v.weight]).to(torch.bfloat16)
    assert torch.equal(wqk, ref_qk) and wv is None
    # release the fp32 masters: the caches keep serving, the parameters are gone
    before = sum(p.numel() for p in eng.parameters())
    rep = eng.prepare(free_masters=True)
    after = sum(p.numel() for p in eng.parameters())
    assert rep["freed_bytes"] > 0 and after < 0.1 * before
    assert torch.equal(blk.attn1.packed_ln(blk.norm1)[0], ref_f) and blk.attn1.to_q.weight.numel() == 0
    assert torch.equal(blk.attn1.packed()[0], ref_qk)            # (the plain layout built above stays cached)
    assert isinstance(blk.norm1, H.LayerNorm) and blk.norm1.weight.numel() > 0        # norms stay (consumed as fp32)
    w, b = eng.model.diffusion_model._emb_pack()
    assert w.shape[0] == sum(rb.out_channels for rb in eng.model.diffusion_model._resblocks)


# ---------------------------------------------------------------------------------- reference YAML / util.py / test.py
def _install_shims():
    """the third-party modules the reference's util.py / test.py import at module level and that do not exist here"""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class OmegaConf:
        load = staticmethod(C.load_config)

    saved = {k: sys.modules.get(k) for k in ("omegaconf", "pytorch_lightning", "torchvision", "torchvision.utils", "dataset",
                                             "dataset.dataloader", "metrics", "util", "test")}
    mod("omegaconf", OmegaConf=OmegaConf, ListConfig=list)
    mod("pytorch_lightning", seed_everything=lambda s: torch.manual_seed(s))
    tv = mod("torchvision")
    tv.utils = mod("torchvision.utils", save_image=lambda *a, **k: None)
    ds = mod("dataset")
    ds.dataloader = mod("dataset.dataloader", get_dataloader=lambda *a, **k: None)     # needs cv2 + datasets: not on the path
    mod("metrics", calc_fid=None, calc_lpips=None)                                      # imports lpips at module level
    return saved


def _restore(saved):
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


@needs_ref
def test_reference_yaml_instantiates_with_reference_keys():
    """configs/test/textdesign_sd_2.yaml (ckpt paths stripped: no checkpoints here) -> the new sgm -> the reference's
    1330 state-dict keys in the reference's order"""
    from sgm.util import instantiate_from_config, skip_param_init
    cfg = C.load_config(os.path.join(REF, "configs/test/textdesign_sd_2.yaml"))
    _strip_ckpt(cfg)
    with skip_param_init():
        eng = instantiate_from_config(cfg.model)
    ref = json.load(open(os.path.join(GOLD, "state_dict_keys.json")))
    sd = eng.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    assert all(list(v.shape) == ref[k] for k, v in sd.items())


@needs_ref
def test_reference_util_and_predict_run_against_the_new_sgm():
    """the reference's own util.py (init_sampling, prepare_batch, deep_copy) and test.py:predict, imported unmodified
    under the shims, drive the new sgm's objects: the sampler they build is the MI355X EulerEDMSampler, and predict's
    calls match the new methods' signatures (autospec mocks stand in for the GPU work)."""
    saved = _install_shims()
    sys.path.insert(0, REF)
    try:
        util = importlib.import_module("util")
        rtest = importlib.import_module("test")
        assert util.__file__.startswith(REF) and rtest.__file__.startswith(REF)
        from sgm.models.diffusion import DiffusionEngine
        from sgm.modules import GeneralConditioner
        from sgm.modules.diffusionmodules.guiders import VanillaCFG
        from sgm.modules.diffusionmodules.sampling import EulerEDMSampler
        assert util.EulerEDMSampler is EulerEDMSampler                        # `from ...sampling import *` resolves to ours

        class _Torch:                                                          # util.py hard-codes torch.device("cuda", ...)
            def __getattr__(self, n):
                return getattr(torch, n)

            @staticmethod
            def device(*a, **k):
                return torch.device("cpu")

        util.torch = _Torch()
        cfgs = C.load_config(os.path.join(REF, "configs/test.yaml"))
        sampler = util.init_sampling(cfgs)
        assert isinstance(sampler, EulerEDMSampler) and isinstance(sampler.guider, VanillaCFG)
        assert sampler.num_steps == cfgs.steps and sampler.guider.scale == cfgs.scale[0] and sampler.s_churn == 0.0
        from udifftext_amd import synth
        batch = synth.synthetic_batch(1, 64, 64, 4, seed=0)
        b, buc = util.prepare_batch(cfgs, batch)
        assert buc["label"] == [""] and torch.equal(b["image"], buc["image"]) and b["image"] is not buc["image"]
        # predict(): conditioner -> get_init_noise -> sampler -> decode, with the reference's keyword arguments
        c, uc = {"concat": torch.zeros(1, 5, 8, 8)}, {"concat": torch.ones(1, 5, 8, 8)}
        x0, z, img = torch.zeros(1, 4, 8, 8), torch.ones(1, 4, 8, 8), torch.full((1, 3, 64, 64), 3.0)
        model = mock.create_autospec(DiffusionEngine, instance=True)
        model.conditioner = mock.create_autospec(GeneralConditioner, instance=True)
        model.conditioner.get_unconditional_conditioning.return_value = (c, uc)
        model.decode_first_stage.return_value = img
        with mock.patch.object(EulerEDMSampler, "get_init_noise", autospec=True, return_value=x0) as gin, \
                mock.patch.object(EulerEDMSampler, "__call__", autospec=True, return_value=z) as call:
            samples, samples_z = rtest.predict(cfgs, model, sampler, batch)
        assert gin.call_count == 1 and call.call_count == 1
        assert call.call_args.kwargs["aae_enabled"] == cfgs.aae_enabled and call.call_args.kwargs["init_step"] == 0
        kw = model.conditioner.get_unconditional_conditioning.call_args.kwargs
        assert kw["force_uc_zero_embeddings"] == cfgs.force_uc_zero_embeddings and kw["batch_uc"]["label"] == [""]
        assert samples_z is z and float(samples.max()) == 1.0                   # clamp((x + 1) / 2, 0, 1)
    finally:
        sys.path.remove(REF)
        _restore(saved)


def test_sd2_inpainting_key_map():
    """the LDM-named SD-2 inpainting checkpoint the reference's training starts from (configs/train.yaml:5): UNet keys load
    straight except attn2 / norm2, the autoencoder loads into first_stage_model only (the LatentEncoder keeps its own file's
    weights, like the reference's strict=False load; mirroring is opt-in), CLIP / EMA / schedule entries drop"""
    from udifftext_amd import ckpt
    ref = json.load(open(os.path.join(GOLD, "state_dict_keys.json")))
    engine_keys = list(ref.keys())
    unet = [k for k in engine_keys if k.startswith("model.diffusion_model.")]
    vae = [k for k in engine_keys if k.startswith("first_stage_model.")]
    sd = {}
    for k in unet:                                         # an SD-2 file has attn2 / norm2 where UDiffText has t_attn / t_norm
        k2 = k.replace(".t_attn.", ".attn2.").replace(".t_norm.", ".norm2.")
        sd[k2] = torch.zeros(1)
    for k in vae:
        sd[k] = torch.ones(1)
    sd.update({"cond_stage_model.model.ln_final.weight": torch.zeros(1), "model_ema.decay": torch.zeros(1), "betas": torch.zeros(1),
               "alphas_cumprod": torch.zeros(1), "sqrt_recip_alphas_cumprod": torch.zeros(1)})
    mapped0, rep0 = ckpt.map_sd2_inpainting(sd, engine_keys)
    assert not rep0["duplicated_to_latent_encoder"] and not any(k.startswith("conditioner.") for k in mapped0)
    assert any(k.startswith("conditioner.embedders.2.model.") for k in rep0["missing"])       # reported missing, as the reference does
    mapped, rep = ckpt.map_sd2_inpainting(sd, engine_keys, mirror_to_latent_encoder=True)
    n_t = sum(1 for k in unet if ".t_attn." in k or ".t_norm." in k)
    assert n_t > 0 and len(rep["dropped_text_cross_attention"]) == n_t
    assert len(rep["loaded"]) == len(unet) - n_t + len(vae)
    assert len(rep["duplicated_to_latent_encoder"]) == len(vae)
    assert all(k in mapped for k in engine_keys if k.startswith("conditioner.embedders.2.model."))
    assert set(rep["dropped_other"]) == {"cond_stage_model.model.ln_final.weight", "model_ema.decay", "betas", "alphas_cumprod",
                                         "sqrt_recip_alphas_cumprod"}
    missing = set(rep["missing"])
    assert all((".t_attn." in k or ".t_norm." in k) for k in missing if k.startswith("model.diffusion_model."))
    assert any(k.startswith("conditioner.embedders.0.") for k in missing)      # the LabelEncoder has its own checkpoint
    assert set(mapped) <= set(engine_keys)


def test_config5_host_logic_fixed_v_multiplier_and_switches(monkeypatch):
    """host side of BASELINE config #5 (no kernel runs): the data-free e4m3 multiplier of a layer's v projection is a power of two
    that puts 448 at >= 12 sigma of the widest output row (+ its constant term) — and never above twice that; the e4m3 attention
    is on only together with the MX8 linears and the LayerNorm-folded GEMMs."""
    import math
    import sgm.modules.hipnn as H
    g = torch.Generator().manual_seed(5)
    for scale, cmax in ((0.02, 0.0), (1.0, 0.3), (7.5, 40.0)):
        w = torch.randn((192, 640), generator=g) * scale
        c = (torch.rand((192,), generator=g) * 2 - 1) * cmax
        m = H.v_fixed_mul(w, c)
        bound = 12.0 * float(w.norm(dim=1).max()) + float(c.abs().max())
        assert m == 2.0 ** round(math.log2(m)) and 224.0 < m * bound <= 448.0, (m, bound)
    assert H.v_fixed_mul(torch.zeros((4, 8)), torch.zeros((4,))) > 0                      # (degenerate weights: finite, positive)
    for fp8, ln, a8, want in ((True, True, True, True), (False, True, True, False), (True, False, True, False), (True, True, False, False)):
        monkeypatch.setattr(H, "FP8_LINEARS", fp8)
        monkeypatch.setattr(H, "LN_GEMM", ln)
        monkeypatch.setattr(H, "FP8_ATTENTION", a8)
        assert H.fp8_attention() is want
        assert H.mx8_width(640) is (fp8 and ln) and H.mx8_width(320) is False
