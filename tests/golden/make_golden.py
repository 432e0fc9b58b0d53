"""Generate the golden fixtures in tests/golden/ by running the REAL reference (read-only at /root/reference)
on its CPU path inside the build container.

The reference's Python cannot travel to the GPU box, so only the resulting vectors (inputs are regenerated
from seeds / the name-keyed synthetic weight recipe, expected outputs are stored) are committed.  Run:

    python tests/golden/make_golden.py            # ~3-4 minutes, needs /root/reference
    python tests/golden/make_golden.py --g11      # only the 50-step trajectory (engine_golden_50.npz), ~3 minutes
    python tests/golden/make_golden.py --g12      # only BASELINE config #2 end to end (engine_golden_512.npz), ~10 minutes
    python tests/golden/make_golden.py --g13      # only the attend-and-excite update (aae_golden.npz), ~2 minutes
    python tests/golden/make_golden.py --g14      # only the training step's loss and parameter gradients (train_golden.npz), ~2 minutes

Import recipe: import transformers first; stub the absent third-party modules
(pytorch_lightning, omegaconf, kornia, open_clip, imageio, seaborn, torchvision, timm); replace xformers'
memory_efficient_attention by exact softmax attention (F.scaled_dot_product_attention); strip every
``ckpt_path`` from configs/test/textdesign_sd_2.yaml; fill the weights with udifftext_amd.synth.
"""
from __future__ import annotations

import json
import os
import sys
import time
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from udifftext_amd import synth  # noqa: E402


def install_stubs():
    import transformers  # noqa: F401  (must come before the torchvision stub)

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _LM(nn.Module):
        def log_dict(self, *a, **k): pass
        def log(self, *a, **k): pass

    mod("pytorch_lightning", LightningModule=_LM, seed_everything=lambda s: torch.manual_seed(s))

    class ListConfig(list):
        pass

    class OmegaConf:
        @staticmethod
        def load(p):
            return yaml.safe_load(open(p))

    mod("omegaconf", ListConfig=ListConfig, OmegaConf=OmegaConf)
    for n in ("kornia", "open_clip", "imageio", "seaborn"):
        mod(n)
    tv = mod("torchvision")
    tv.utils = mod("torchvision.utils", save_image=lambda *a, **k: None)
    tr = mod("torchvision.transforms", Compose=object, Resize=object, Normalize=object, Grayscale=object,
             InterpolationMode=types.SimpleNamespace(BICUBIC=3))
    tv.transforms = tr
    timm = mod("timm")
    timm.models = mod("timm.models")
    timm.models.vision_transformer = mod("timm.models.vision_transformer", VisionTransformer=nn.Module)


def import_reference():
    install_stubs()
    sys.path.insert(0, REF)
    import sgm  # noqa: F401
    import sgm.modules.attention as A
    import sgm.modules.diffusionmodules.model as Mo

    def mea(q, k, v, attn_bias=None, op=None):
        return F.scaled_dot_product_attention(q, k, v)

    shim = types.SimpleNamespace(ops=types.SimpleNamespace(memory_efficient_attention=mea))
    A.xformers = shim
    Mo.xformers = shim
    return sgm


def strip_ckpt(cfg):
    if isinstance(cfg, dict):
        cfg.pop("ckpt_path", None)
        for v in cfg.values():
            strip_ckpt(v)
    elif isinstance(cfg, list):
        for v in cfg:
            strip_ckpt(v)


def sub(t: torch.Tensor, n: int = 4096) -> np.ndarray:
    """strided sub-sample of a flattened tensor (deterministic, <= n values)"""
    f = t.detach().float().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().copy()


def stats(t: torch.Tensor) -> np.ndarray:
    f = t.detach().double()
    return np.array([f.sum().item(), f.abs().sum().item(), (f * f).sum().item()], dtype=np.float64)


def g12(model, S, t0):
    """G12 — BASELINE config #2 end to end on ONE image: 512x512, the 9-character label "Diffusion", 50 Euler steps
    through the real EulerEDMSampler.__call__ (reference sampling.py:355-420), CFG 5, noise_iters 0; conditioning from the
    real GeneralConditioner under torch.manual_seed(1234) (draw order pinned), x0 under torch.manual_seed(512).  Stored:
    conditioning sub-samples, x0, the latent after steps 10 / 25 / 50 and a sub-sample of the decoded image."""
    import io, contextlib
    batch = synth.synthetic_batch(1, 512, 512, 9, seed=12)
    torch.manual_seed(1234)
    buc = {k: (v.clone() if isinstance(v, torch.Tensor) else list(v)) for k, v in batch.items()}
    buc["label"] = ["" for _ in batch["label"]]
    buc["txt"] = ["" for _ in batch["txt"]]
    c, uc = model.conditioner.get_unconditional_conditioning(batch, batch_uc=buc, force_uc_zero_embeddings=["label"])
    print(f"[golden] G12 conditioning done ({time.time() - t0:.1f}s)")
    sampler = S.EulerEDMSampler(
        num_steps=50,
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"},
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 5.0}},
        s_churn=0.0, s_tmin=0.0, s_tmax=999.0, s_noise=1.0, verbose=False, device="cpu")
    traj = []
    orig_step = sampler.sampler_step

    def recording_step(*a, **k):
        r = orig_step(*a, **k)
        traj.append(r[0].clone())
        if len(traj) % 10 == 0:
            print(f"[golden] G12 step {len(traj)} ({time.time() - t0:.1f}s)", flush=True)
        return r

    sampler.sampler_step = recording_step
    cfgs = types.SimpleNamespace(batch_size=1, channel=4, factor=8, gpu=0, noise_iters=0)
    torch.manual_seed(512)
    x0 = sampler.get_init_noise(cfgs, model, cond=c, batch=batch, uc=uc)
    with contextlib.redirect_stdout(io.StringIO()):
        z50 = sampler(model, x0.clone(), cond=c, batch=batch, uc=uc, init_step=0, aae_enabled=False, detailed=False)
    assert len(traj) == 50 and torch.equal(traj[-1], z50)
    dec = model.decode_first_stage(z50)
    out = {"g12_c_concat": c["concat"].numpy(), "g12_uc_concat": uc["concat"].numpy(),
           "g12_c_txt_sub": c["t_crossattn"][:, :, ::16].numpy(), "g12_x0": x0.numpy(),
           "g12_latent_10": traj[9].numpy(), "g12_latent_25": traj[24].numpy(), "g12_latent_50": z50.numpy(),
           "g12_decoded_sub": dec[:, :, ::8, ::8].numpy(),
           "g12_latent_rms": np.array([t.pow(2).mean().sqrt().item() for t in traj])}
    np.savez_compressed(os.path.join(HERE, "engine_golden_512.npz"), **out)
    print(f"[golden] G12 512x512 50-step trajectory done ({time.time() - t0:.1f}s)")


sys.path.insert(0, os.path.dirname(HERE))
from aae_fixture import aae_batch, aae_functional_weights, train_batch, sub as gsub  # noqa: E402


def g13(model, S, t0):
    """G13 — attend-and-excite: ONE update of the real EulerEDMSampler.attend_and_excite (reference sampling.py:233-252:
    x <- x - alpha * d local_loss / d x, autograd through the whole UNet from the t_attn probability maps of the 16x16 level) on a
    128x128 image (16x16 latents, B = 1 — the reference's implicit-gradient call only works for one sample), at the sampler's step 2
    of 10, on ``aae_batch()``.  Stored: x, sigma, alpha, the updated x, the local loss and the gradient torch.autograd.grad returned
    inside that call.
    G13s — the same reverse pass with a DENSE cotangent on every counted map: the real reference network under autograd, scalarised by
    the smooth functional sum_k <R_k, attn_map_k> / count (R_k fixed pseudo-random, ``aae_functional_weights``) instead of the hard
    min / max loss: d / d x of it."""
    batch = aae_batch()
    torch.manual_seed(1234)
    buc = {k: (v.clone() if isinstance(v, torch.Tensor) else list(v)) for k, v in batch.items()}
    buc["label"] = ["" for _ in batch["label"]]
    buc["txt"] = ["" for _ in batch["txt"]]
    c, uc = model.conditioner.get_unconditional_conditioning(batch, batch_uc=buc, force_uc_zero_embeddings=["label"])
    sampler = S.EulerEDMSampler(
        num_steps=10,
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"},
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 5.0}},
        s_churn=0.0, s_tmin=0.0, s_tmax=999.0, s_noise=1.0, verbose=False, device="cpu")
    sigmas = sampler.discretization(10, device="cpu")
    i = 2
    g = torch.Generator().manual_seed(13)
    x = torch.randn((1, 4, 16, 16), generator=g) * float(sigmas[i])
    sigma = torch.ones([1]) * sigmas[i]
    alpha = float(20 * np.sqrt(np.linspace(1.0, 0, 11)[i]))
    grads = []
    real_grad = torch.autograd.grad

    def recording_grad(*a, **k):          # (x - x') / alpha in fp32 would carry ~2 % of cancellation noise: record the gradient itself
        r = real_grad(*a, **k)
        grads.append(r[0].detach().clone())
        return r

    unet = model.model.diffusion_model
    with torch.enable_grad():
        torch.autograd.grad = recording_grad
        try:
            x2 = sampler.attend_and_excite(x, model, sigma, c, batch, alpha, False, 0.0)
        finally:
            torch.autograd.grad = real_grad
        # the loss value of that call (attend_and_excite does not return it) and the selection margins: the same forward once more
        c_noise = sampler.get_c_noise(x, model, sigma)
        xg = x.clone().requires_grad_(True)
        model.model(xg, c_noise, c)
        ll = model.loss_fn.get_min_local_loss(unet.attn_map_cache, batch["mask"], batch["seg_mask"])
        used = [it for it in unet.attn_map_cache if it["name"].endswith("t_attn") and it["size"] >= model.loss_fn.min_attn_size]
        smooth = sum((aae_functional_weights(it["attn_map"].shape, k) * it["attn_map"]).sum() for k, it in enumerate(used)) / len(used)
        (gs,) = real_grad(smooth, [xg])
    assert len(grads) == 1
    grad = grads[0]
    assert torch.allclose(x - alpha * grad, x2.detach(), rtol=0, atol=1e-6)
    out = {"g13_x": x.numpy(), "g13_sigma": sigma.numpy(), "g13_alpha": np.array([alpha]), "g13_c_noise": c_noise.numpy(),
           "g13_x_updated": x2.detach().numpy(), "g13_local_loss": ll.detach().numpy(), "g13_grad": grad.numpy(),
           "g13_c_concat": c["concat"].numpy(), "g13_c_txt": c["t_crossattn"].numpy(),
           "g13s_value": np.array([float(smooth)]), "g13s_grad": gs.numpy(),
           "g13_map_names": np.array([it["name"] for it in used])}
    np.savez_compressed(os.path.join(HERE, "aae_golden.npz"), **out)
    print(f"[golden] G13 attend-and-excite update done ({time.time() - t0:.1f}s): loss {ll.item():.6f}, |grad| rms "
          f"{grad.pow(2).mean().sqrt().item():.3e}, alpha {alpha:.3f}; smooth functional {float(smooth):.6f}, |grad| rms "
          f"{gs.pow(2).mean().sqrt().item():.3e}; maps {[it['name'] for it in used]}")


def g14(model, t0):
    """G14 — one training step's loss and parameter gradients (second half): the real FullLoss.__call__ (reference
    loss.py:131-176) under torch.autograd on ``train_batch()`` (B = 2, 16x16 latents), gradients with respect to the parameters
    DiffusionEngine.configure_optimizers selects (diffusion.py:202-217: opt_keys t_attn, t_norm — 112 tensors, 75.9 M values).
    The step's random draws are recorded (conditioning incl. its ucg draw, sigma indices, noise).  Stored: the loss dict, and for every
    trained tensor [sum, sum |.|, sum squares] + a strided sub-sample of (a) the gradient of loss/full_loss, (b) the gradient with
    lambda_local_loss = 0 (the eps-prediction loss alone: a smooth function of the parameters — the hard arg-max selections of
    get_local_loss sit on near-uniform maps and are not reproducible to the last bit across hosts), (c) G14s: the gradient of the
    smooth functional sum_k <R_k, attn_map_k> / count of the same forward."""
    batch = train_batch()
    torch.manual_seed(4321)
    z = torch.randn((2, 4, 16, 16)) * 0.8
    sigma_idx = torch.tensor([700, 250])
    noise = torch.randn((2, 4, 16, 16))
    loss_fn = model.loss_fn
    unet = model.model.diffusion_model
    names = [("model." + n) for n, _ in model.model.named_parameters() if any(k in n for k in ("t_attn", "t_norm"))]
    params = [p for n, p in model.model.named_parameters() if any(k in n for k in ("t_attn", "t_norm"))]
    recorded = {}
    real_cond = model.conditioner.forward

    def cond_once(b, *a, **k):
        if "cond" not in recorded:
            recorded["cond"] = real_cond(b, *a, **k)
        return recorded["cond"]

    real_randn_like = torch.randn_like
    loss_fn.sigma_sampler = lambda n, rand=None: loss_fn_sigmas[sigma_idx]
    loss_fn_sigmas = model.denoiser.sigmas
    out = {}
    try:
        model.conditioner.forward = cond_once
        torch.randn_like = lambda t, **k: noise.clone()
        for p_ in params:
            p_.requires_grad_(True)
        with torch.enable_grad():
            for tag, lam in (("full", 0.01), ("diff", 0.0)):
                loss_fn.lambda_local_loss = lam
                loss, ld = loss_fn(model.model, model.denoiser, model.conditioner, z, batch, model.first_stage_model, model.scale_factor)
                gs = torch.autograd.grad(loss, params)
                if tag == "full":
                    for k, v in ld.items():
                        out["g14_" + k.replace("/", "_")] = np.array([float(v)])
                out[f"g14_{tag}_stats"] = np.stack([stats(g) for g in gs])
                out[f"g14_{tag}_sub"] = np.stack([np.pad(gsub(g).numpy(), (0, 512 - gsub(g).numel())) for g in gs])
            # G14s: dense cotangents on the counted maps of one more (identical) forward
            loss_fn.lambda_local_loss = 0.01
            cond = recorded["cond"]
            sig = loss_fn_sigmas[sigma_idx]
            noised = z + noise * sig[:, None, None, None]
            model.denoiser(model.model, noised, sig, cond)
            used = [it for it in unet.attn_map_cache if it["name"].endswith("t_attn") and it["size"] >= loss_fn.min_attn_size]
            smooth = sum((aae_functional_weights(it["attn_map"].shape, k) * it["attn_map"]).sum() for k, it in enumerate(used)) / len(used)
            gss = torch.autograd.grad(smooth, params, allow_unused=True)
            gss = [g if g is not None else torch.zeros_like(p_) for g, p_ in zip(gss, params)]
            out["g14s_value"] = np.array([float(smooth)])
            out["g14s_stats"] = np.stack([stats(g) for g in gss])
            out["g14s_sub"] = np.stack([np.pad(gsub(g).numpy(), (0, 512 - gsub(g).numel())) for g in gss])
    finally:
        model.conditioner.forward = real_cond
        torch.randn_like = real_randn_like
        for p_ in params:
            p_.requires_grad_(False)
    cond = recorded["cond"]
    out.update({"g14_z": z.numpy(), "g14_sigma_idx": sigma_idx.numpy(), "g14_noise": noise.numpy(), "g14_c_concat": cond["concat"].detach().numpy(),
                "g14_c_txt": cond["t_crossattn"].detach().numpy(), "g14_names": np.array(names)})
    np.savez_compressed(os.path.join(HERE, "train_golden.npz"), **out)
    print(f"[golden] G14 training step done ({time.time() - t0:.1f}s): " + ", ".join(f"{k} {float(out[k][0]):.6f}" for k in out if k.startswith("g14_loss"))
          + f"; {len(names)} trained tensors, |grad| rms full {np.sqrt(out['g14_full_stats'][:, 2].sum() / sum(p_.numel() for p_ in params)):.3e}")


def main(only_g11: bool = False, only_g12: bool = False, only_g13: bool = False, only_g14: bool = False):
    t0 = time.time()
    torch.set_grad_enabled(False)
    import_reference()
    from sgm.util import instantiate_from_config
    out = {}

    # ------------------------------------------------------------------ engine with synthetic weights
    cfg = yaml.safe_load(open(os.path.join(REF, "configs/test/textdesign_sd_2.yaml")))
    strip_ckpt(cfg)
    model = instantiate_from_config(cfg["model"]).eval()
    print(f"[golden] reference engine built in {time.time() - t0:.1f}s")
    # REFERENCE QUIRK (documented in DESIGN.md): GeneralConditioner installs ``embedder.train = disabled_train``
    # on a LabelEncoder that was constructed in training mode (encoders/modules.py:110-124), so no later
    # ``.eval()`` reaches it and its Dropout(0.1) layers stay ACTIVE at inference — the reference's label
    # embedding is a random variable.  The goldens pin the deterministic (dropout-free) network: force eval.
    le_ref = model.conditioner.embedders[0]
    out_quirk = np.array([int(le_ref.training)])
    nn.Module.train(le_ref, False)
    assert not le_ref.training and not le_ref.encoder.layers[0].dropout.training
    sdict = model.state_dict()
    keys = {k: list(v.shape) for k, v in sdict.items()}
    json.dump(keys, open(os.path.join(HERE, "state_dict_keys.json"), "w"), indent=0)
    for name, p in sdict.items():
        if synth.is_computed_buffer(name):
            continue
        p.copy_(synth.synthetic_tensor(name, tuple(p.shape)))
    print(f"[golden] {len(keys)} state-dict entries filled ({time.time() - t0:.1f}s)")

    # ------------------------------------------------------------------ G1 sigma tables / quantisation
    from sgm.modules.diffusionmodules.discretizer import LegacyDDPMDiscretization
    disc = LegacyDDPMDiscretization()
    for n in (2, 10, 50):
        out[f"g1_sigmas_{n}"] = disc(n).numpy()
    out["g1_denoiser_sigmas"] = model.denoiser.sigmas.numpy()
    s50 = disc(50)[:-1]
    out["g1_cnoise_50"] = model.denoiser.possibly_quantize_c_noise(model.denoiser.possibly_quantize_sigma(s50)).numpy()
    out["g1_gkernel"] = model.loss_fn.g_kernel.numpy()

    # ------------------------------------------------------------------ G2 timestep embedding
    from sgm.modules.diffusionmodules.util import timestep_embedding
    out["g2_temb"] = timestep_embedding(torch.tensor([999, 979, 19, 0]), 320).numpy()

    # ------------------------------------------------------------------ G3 LabelEncoder
    le = model.conditioner.embedders[0]
    labels = ["TEXT", "Diffusion", "MI355XNative", "Te9~ é"]
    out["g3_index"] = le.get_index(labels).numpy()
    out["g3_reference_label_encoder_left_in_training_mode"] = out_quirk
    emb = le(labels)
    out["g3_label_sub"] = emb[:, :, ::16].numpy()
    out["g3_label_stats"] = stats(emb)
    out["g3_pe"] = le.pos_embedding.pe[:, ::64].numpy()

    # ------------------------------------------------------------------ G4 SpatialRescaler
    batch256 = synth.synthetic_batch(1, 256, 256, 4, seed=0)
    out["g4_mask_ds"] = model.conditioner.embedders[1](batch256["mask"]).numpy()

    # ------------------------------------------------------------------ G5 VAE on a 64x64 image
    g = torch.Generator().manual_seed(5)
    img64 = torch.rand((1, 3, 64, 64), generator=g) * 2 - 1
    fs = model.first_stage_model
    mom = fs.quant_conv(fs.encoder(img64))
    out["g5_moments"] = mom.numpy()
    z8 = torch.randn((1, 4, 8, 8), generator=g) * 3.0
    out["g5_decoded"] = fs.decode(z8).numpy()

    # ------------------------------------------------------------------ G6 conditioner (draw order pinned by seed)
    torch.manual_seed(1234)
    buc = {k: (v.clone() if isinstance(v, torch.Tensor) else list(v)) for k, v in batch256.items()}
    buc["label"] = ["" for _ in batch256["label"]]
    buc["txt"] = ["" for _ in batch256["txt"]]
    c, uc = model.conditioner.get_unconditional_conditioning(batch256, batch_uc=buc, force_uc_zero_embeddings=["label"])
    out["g6_c_concat"] = c["concat"].numpy()
    out["g6_uc_concat"] = uc["concat"].numpy()
    out["g6_c_txt_sub"] = c["t_crossattn"][:, :, ::16].numpy()
    out["g6_uc_txt_absmax"] = np.array([uc["t_crossattn"].abs().max().item()])

    # ------------------------------------------------------------------ G7 one UNet call (CFG pair) + per-block taps
    unet = model.model.diffusion_model
    taps = {}
    hooks = []
    for i, blk in enumerate(unet.input_blocks):
        hooks.append(blk.register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(f"input_blocks.{i}", o)))
    hooks.append(unet.middle_block.register_forward_hook(lambda m, a, o: taps.__setitem__("middle_block", o)))
    for i, blk in enumerate(unet.output_blocks):
        hooks.append(blk.register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(f"output_blocks.{i}", o)))
    g = torch.Generator().manual_seed(7)
    x7 = torch.randn((1, 4, 32, 32), generator=g)
    xin = torch.cat([torch.cat([x7, x7]), torch.cat([uc["concat"], c["concat"]])], dim=1)
    tctx = torch.cat([uc["t_crossattn"], c["t_crossattn"]])
    ts = torch.tensor([999, 999])
    eps = unet(xin, timesteps=ts, t_context=tctx)
    for h in hooks:
        h.remove()
    out["g7_x"] = x7.numpy()
    out["g7_eps"] = eps.numpy()
    for k, v in taps.items():
        out[f"g7_tap_{k}_sub"] = sub(v, 2048)
        out[f"g7_tap_{k}_stats"] = stats(v)
    names = []
    for item in unet.attn_map_cache:
        names.append([item["name"], int(item["heads"]), int(item["size"]), list(item["attn_map"].shape)])
        out[f"g7_attn_{item['name']}_sub"] = sub(item["attn_map"], 2048)
    json.dump(names, open(os.path.join(HERE, "attn_map_names.json"), "w"))
    # ------------------------------------------------------------------ G8 local loss on those maps
    ll = model.loss_fn.get_min_local_loss(unet.attn_map_cache, batch256["mask"], batch256["seg_mask"])
    out["g8_local_loss"] = ll.numpy()
    print(f"[golden] G1-G8 done ({time.time() - t0:.1f}s)")

    # ------------------------------------------------------------------ G9 config #1: 256^2, 10 steps, "TEXT", B=1
    import sgm.modules.diffusionmodules.sampling as S

    class _TorchProxy:
        def __getattr__(self, n):
            return getattr(torch, n)

        @staticmethod
        def device(*a, **k):
            return torch.device("cpu")

    S.torch = _TorchProxy()
    if only_g12:
        g12(model, S, t0)
        return
    if only_g13:
        g13(model, S, t0)
        return
    if only_g14:
        g14(model, t0)
        return
    sampler = S.EulerEDMSampler(
        num_steps=10,
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"},
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 5.0}},
        s_churn=0.0, s_tmin=0.0, s_tmax=999.0, s_noise=1.0, verbose=False, device="cpu")
    cfgs = types.SimpleNamespace(batch_size=1, channel=4, factor=8, gpu=0, noise_iters=0)
    torch.manual_seed(99)
    x0 = sampler.get_init_noise(cfgs, model, cond=c, batch=batch256, uc=uc)
    out["g9_x0"] = x0.numpy()
    z = sampler(model, x0.clone(), cond=c, batch=batch256, uc=uc, init_step=0, aae_enabled=False, detailed=False)
    out["g9_latent"] = z.numpy()
    dec = model.decode_first_stage(z)
    samples = torch.clamp((dec + 1.0) / 2.0, min=0.0, max=1.0)
    out["g9_decoded_sub"] = dec[:, :, ::8, ::8].numpy()
    out["g9_samples_stats"] = stats(samples)
    print(f"[golden] G9 10-step trajectory done ({time.time() - t0:.1f}s)")

    # noise search (noise_iters = 2  -> 3 CPU draws, 4 UNet calls)
    import io, contextlib
    if not only_g11:
        cfgs.noise_iters = 2
        torch.manual_seed(77)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            xs = sampler.get_init_noise(cfgs, model, cond=c, batch=batch256, uc=uc)
        out["g9_search_x0"] = xs.numpy()
        line = [l for l in buf.getvalue().splitlines() if l.startswith("Init local loss")][0]
        best, worst = float(line.split("Best")[1].split("Worst")[0]), float(line.split("Worst")[1])
        out["g9_search_scores"] = np.array([best, worst])
        print(f"[golden] noise search done ({time.time() - t0:.1f}s)")

    if not only_g11:
        np.savez_compressed(os.path.join(HERE, "engine_golden.npz"), **out)

    # ------------------------------------------------------------------ G11 the benchmarked step count: 50 steps
    # (configs/test.yaml:20 ``steps: 50``; reference sampling.py:355-420 at num_steps=50), 256x256, "TEXT", B=1, CFG 5.
    # The latent after steps 10 / 25 / 50 is stored (a wrapper around sampler_step records the loop's x), so the GPU test
    # can state a tolerance per horizon: with random weights the trajectory is chaotic and bf16 error grows with depth.
    sampler50 = S.EulerEDMSampler(
        num_steps=50,
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"},
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 5.0}},
        s_churn=0.0, s_tmin=0.0, s_tmax=999.0, s_noise=1.0, verbose=False, device="cpu")
    traj = []
    orig_step = sampler50.sampler_step

    def recording_step(*a, **k):
        r = orig_step(*a, **k)
        traj.append(r[0].clone())
        return r

    sampler50.sampler_step = recording_step
    cfgs.noise_iters = 0
    torch.manual_seed(4242)
    x0 = sampler50.get_init_noise(cfgs, model, cond=c, batch=batch256, uc=uc)
    with contextlib.redirect_stdout(io.StringIO()):
        z50 = sampler50(model, x0.clone(), cond=c, batch=batch256, uc=uc, init_step=0, aae_enabled=False, detailed=False)
    assert len(traj) == 50 and torch.equal(traj[-1], z50)
    g11 = {"g11_x0": x0.numpy(), "g11_latent_10": traj[9].numpy(), "g11_latent_25": traj[24].numpy(),
           "g11_latent_50": z50.numpy(), "g11_decoded_sub": model.decode_first_stage(z50)[:, :, ::8, ::8].numpy(),
           "g11_latent_rms": np.array([t.pow(2).mean().sqrt().item() for t in traj])}
    np.savez_compressed(os.path.join(HERE, "engine_golden_50.npz"), **g11)
    print(f"[golden] G11 50-step trajectory done ({time.time() - t0:.1f}s)")
    if only_g11:
        return

    # ------------------------------------------------------------------ G10 reference MODULES at small shapes
    mods = {}
    from sgm.modules.attention import BasicTransformerBlock, CrossAttention, FeedForward, MemoryEfficientCrossAttention, SpatialTransformer
    from sgm.modules.diffusionmodules.model import Downsample as VDown, MemoryEfficientAttnBlock, ResnetBlock, Upsample as VUp
    from sgm.modules.diffusionmodules.openaimodel import Downsample, ResBlock, Upsample

    def build(name, m):
        m = m.eval()
        synth.fill_module_(m, prefix=f"g10.{name}.")
        return m

    g = torch.Generator().manual_seed(10)
    rn = lambda *s: torch.randn(s, generator=g)
    x = rn(2, 64, 8, 8); emb = rn(2, 256)
    mods["res_64_128"] = build("res_64_128", ResBlock(64, 256, 0.0, out_channels=128))(x, emb)
    mods["res_64_64"] = build("res_64_64", ResBlock(64, 256, 0.0, out_channels=64))(x, emb)
    mods["in_res_x"], mods["in_res_emb"] = x, emb
    mods["up_64"] = build("up_64", Upsample(64, True))(x)
    mods["down_64"] = build("down_64", Downsample(64, True))(x)
    t = rn(2, 64, 128); ctx = rn(2, 12, 96)
    mods["in_tokens"], mods["in_ctx"] = t, ctx
    mods["selfattn_128"] = build("selfattn_128", MemoryEfficientCrossAttention(128, heads=2, dim_head=64))(t)
    ca = build("xattn_128", CrossAttention(128, context_dim=96, heads=2, dim_head=64))
    ca.attn_map_cache = {"size": None, "attn_map": None}
    mods["xattn_128"] = ca(t, context=ctx)
    mods["xattn_128_map"] = ca.attn_map_cache["attn_map"]
    mods["xattn_128_single"] = ca(t, context=ctx[:, :1])
    mods["ff_128"] = build("ff_128", FeedForward(128, glu=True))(t)
    mods["block_128"] = build("block_128", BasicTransformerBlock(128, 2, 64, t_context_dim=96))(t, t_context=ctx)
    st = build("st_128", SpatialTransformer(128, 2, 64, depth=1, t_context_dim=96, use_linear=True))
    xs_ = rn(2, 128, 8, 8)
    mods["in_st_x"] = xs_
    mods["st_128"] = st(xs_, t_context=ctx)
    xv = rn(2, 64, 8, 8)
    mods["in_vae_x"] = xv
    mods["vres_64_128"] = build("vres_64_128", ResnetBlock(in_channels=64, out_channels=128, dropout=0.0, temb_channels=0))(xv, None)
    mods["vres_64_64"] = build("vres_64_64", ResnetBlock(in_channels=64, out_channels=64, dropout=0.0, temb_channels=0))(xv, None)
    mods["vattn_64"] = build("vattn_64", MemoryEfficientAttnBlock(64))(xv)
    mods["vdown_64"] = build("vdown_64", VDown(64, True))(xv)
    mods["vup_64"] = build("vup_64", VUp(64, True))(xv)
    np.savez_compressed(os.path.join(HERE, "module_golden.npz"), **{k: v.numpy() for k, v in mods.items()})
    print(f"[golden] all fixtures written ({time.time() - t0:.1f}s)")


if __name__ == "__main__":
    if "--g13" in sys.argv:
        main(only_g13=True)
        sys.exit(0)
    if "--g14" in sys.argv:
        main(only_g14=True)
        sys.exit(0)
    main(only_g11="--g11" in sys.argv, only_g12="--g12" in sys.argv)
    if "--all" in sys.argv:
        main(only_g12=True)
