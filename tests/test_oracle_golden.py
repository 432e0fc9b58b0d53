"""Pins the CPU oracle (oracle/) against golden vectors captured from the REAL reference
(tests/golden/make_golden.py; fixtures engine_golden.npz / module_golden.npz / state_dict_keys.json).
CPU-only; runs in the build container and on the GPU box alike (reads nothing outside the repo)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import nets, sampling, spec
from udifftext_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = dict(rtol=2e-4, atol=2e-4)     # fp32 vs fp32, different association order


@pytest.fixture(scope="module")
def eg():
    return np.load(os.path.join(GOLD, "engine_golden.npz"))


@pytest.fixture(scope="module")
def mg():
    return np.load(os.path.join(GOLD, "module_golden.npz"))


@pytest.fixture(scope="module")
def cfg():
    return spec.EngineConfig()


@pytest.fixture(scope="module")
def sd(cfg):
    """full synthetic state dict of the engine (1.36 B parameters, name-keyed recipe)"""
    torch.set_grad_enabled(False)
    d = synth.synthetic_state_dict(spec.engine_param_shapes(cfg))
    d["denoiser.sigmas"] = sampling.denoiser_sigma_table(1000)
    d["loss_fn.g_kernel"] = sampling.gaussian_kernel(3, 1.0, 12)
    d["conditioner.embedders.0.pos_embedding.pe"] = nets.positional_encoding(12, 2048)
    return d


def _sub(t, n):
    f = t.detach().float().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy()


def _stats(t):
    f = t.detach().double()
    return np.array([f.sum().item(), f.abs().sum().item(), (f * f).sum().item()])


def test_state_dict_names_and_shapes(cfg):
    ref = json.load(open(os.path.join(GOLD, "state_dict_keys.json")))
    mine = spec.engine_param_shapes(cfg)
    assert len(ref) == 1330 and len(mine) == 1330
    assert [n for n, _ in mine] == list(ref.keys())          # same order as the reference's state_dict()
    for n, s in mine:
        assert list(s) == ref[n], n


def test_g1_sigma_tables(eg, sd):
    for n in (2, 10, 50):
        np.testing.assert_array_equal(sampling.ddpm_sigmas(n).numpy(), eg[f"g1_sigmas_{n}"])
    np.testing.assert_array_equal(sd["denoiser.sigmas"].numpy(), eg["g1_denoiser_sigmas"])
    s50 = sampling.ddpm_sigmas(50)[:-1]
    idx = sampling.sigma_to_idx(sd["denoiser.sigmas"], s50)
    np.testing.assert_array_equal(idx.numpy(), eg["g1_cnoise_50"])
    assert idx[0].item() == 999 and idx[-1].item() == 19
    np.testing.assert_allclose(sd["loss_fn.g_kernel"].numpy(), eg["g1_gkernel"], rtol=1e-6)


def test_g2_timestep_embedding(eg):
    got = nets.timestep_embedding(torch.tensor([999, 979, 19, 0]), 320)
    np.testing.assert_allclose(got.numpy(), eg["g2_temb"], rtol=1e-6, atol=1e-6)


def test_g3_label_encoder(eg, sd, cfg):
    labels = ["TEXT", "Diffusion", "MI355XNative", "Te9~ é"]
    np.testing.assert_array_equal(nets.label_indices(labels).numpy(), eg["g3_index"])
    np.testing.assert_allclose(sd["conditioner.embedders.0.pos_embedding.pe"][:, ::64].numpy(), eg["g3_pe"], atol=1e-6)
    emb = nets.label_encoder(sd, labels, cfg.label)
    np.testing.assert_allclose(emb[:, :, ::16].numpy(), eg["g3_label_sub"], **TOL)
    np.testing.assert_allclose(_stats(emb), eg["g3_label_stats"], rtol=1e-4)
    with pytest.raises(AssertionError):
        nets.label_indices(["x" * 13])


def test_g4_mask_rescale(eg):
    import torch.nn.functional as F
    b = synth.synthetic_batch(1, 256, 256, 4, seed=0)
    got = F.interpolate(b["mask"], scale_factor=0.125, mode="bilinear")
    np.testing.assert_allclose(got.numpy(), eg["g4_mask_ds"], atol=1e-6)
    m = b["mask"]
    centre = 0.25 * (m[..., 3::8, 3::8] + m[..., 3::8, 4::8] + m[..., 4::8, 3::8] + m[..., 4::8, 4::8])
    np.testing.assert_allclose(centre.numpy(), eg["g4_mask_ds"], atol=1e-6)      # the shortcut the HIP kernel uses


def test_g5_vae(eg, sd, cfg):
    g = torch.Generator().manual_seed(5)
    img = torch.rand((1, 3, 64, 64), generator=g) * 2 - 1
    mom = nets.vae_encode_moments(sd, img, cfg.vae)
    np.testing.assert_allclose(mom.numpy(), eg["g5_moments"], **TOL)
    z8 = torch.randn((1, 4, 8, 8), generator=g) * 3.0
    dec = nets.vae_decode(sd, z8, cfg.vae)
    np.testing.assert_allclose(dec.numpy(), eg["g5_decoded"], **TOL)


@pytest.fixture(scope="module")
def cond256(sd, cfg):
    batch = synth.synthetic_batch(1, 256, 256, 4, seed=0)
    torch.manual_seed(1234)
    c, uc = sampling.conditioning(sd, cfg, batch)
    return batch, c, uc


def test_g6_conditioner(eg, cond256):
    _, c, uc = cond256
    np.testing.assert_allclose(c["concat"].numpy(), eg["g6_c_concat"], **TOL)
    np.testing.assert_allclose(uc["concat"].numpy(), eg["g6_uc_concat"], **TOL)
    np.testing.assert_allclose(c["t_crossattn"][:, :, ::16].numpy(), eg["g6_c_txt_sub"], **TOL)
    assert uc["t_crossattn"].abs().max().item() == 0.0 == eg["g6_uc_txt_absmax"][0]
    # c and uc see different posterior noise (RNG contract)
    assert not np.allclose(c["concat"][:, 1:].numpy(), uc["concat"][:, 1:].numpy())


def test_g7_g8_unet_call(eg, sd, cfg, cond256):
    batch, c, uc = cond256
    x7 = torch.from_numpy(eg["g7_x"])
    xin = torch.cat([torch.cat([x7, x7]), torch.cat([uc["concat"], c["concat"]])], dim=1)
    tctx = torch.cat([uc["t_crossattn"], c["t_crossattn"]])
    maps, taps = [], {}
    eps = nets.unet_forward(sd, xin, torch.tensor([999, 999]), tctx, cfg.unet, attn_maps=maps, taps=taps)
    np.testing.assert_allclose(eps.numpy(), eg["g7_eps"], rtol=1e-3, atol=1e-3)
    for k, v in taps.items():
        np.testing.assert_allclose(_sub(v, 2048), eg[f"g7_tap_{k}_sub"], rtol=1e-3, atol=1e-3, err_msg=k)
        np.testing.assert_allclose(_stats(v), eg[f"g7_tap_{k}_stats"], rtol=1e-3, err_msg=k)
    names = json.load(open(os.path.join(GOLD, "attn_map_names.json")))
    assert len(maps) == len(names) == 16
    for item, (name, heads, size, shape) in zip(maps, names):
        assert item["name"] == name and item["heads"] == heads and item["size"] == size
        assert list(item["attn_map"].shape) == shape
        np.testing.assert_allclose(_sub(item["attn_map"], 2048), eg[f"g7_attn_{name}_sub"], rtol=1e-3, atol=1e-5)
    ll = sampling.min_local_loss(maps, batch["mask"], batch["seg_mask"], sd["loss_fn.g_kernel"])
    np.testing.assert_allclose(ll.numpy(), eg["g8_local_loss"], rtol=1e-3, atol=1e-5)
    # uncond half: zero context -> K = V = 0 -> uniform probabilities
    assert torch.allclose(maps[0]["attn_map"][:maps[0]["heads"]], torch.full((1,), 1.0 / 12))


def test_g9_ten_step_trajectory_and_noise_search(eg, sd, cfg, cond256):
    batch, c, uc = cond256
    torch.manual_seed(99)
    x0, _ = sampling.get_init_noise(sd, cfg, (1, 4, 32, 32), c, uc, batch, 0, 5.0)
    np.testing.assert_array_equal(x0.numpy(), eg["g9_x0"])
    z = sampling.euler_sample(sd, cfg, x0.clone(), c, uc, 10, 5.0)
    # 10 chaotic steps with random weights: fp32 association differences grow; latent scale is ~15
    np.testing.assert_allclose(z.numpy(), eg["g9_latent"], rtol=5e-3, atol=5e-2)
    dec = nets.vae_decode(sd, z / cfg.scale_factor, cfg.vae)
    np.testing.assert_allclose(dec[:, :, ::8, ::8].numpy(), eg["g9_decoded_sub"], rtol=2e-2, atol=2e-2)
    samples = torch.clamp((dec + 1) / 2, 0, 1)
    np.testing.assert_allclose(_stats(samples), eg["g9_samples_stats"], rtol=1e-2)
    torch.manual_seed(77)
    xs, scores = sampling.get_init_noise(sd, cfg, (1, 4, 32, 32), c, uc, batch, 2, 5.0)
    np.testing.assert_array_equal(xs.numpy(), eg["g9_search_x0"])
    np.testing.assert_allclose([min(scores), max(scores)], eg["g9_search_scores"], rtol=1e-3, atol=1e-5)


# ------------------------------------------------------------------------------ reference MODULES (G10)
def _msd(name, shapes):
    return synth.synthetic_state_dict([(f"g10.{name}.{k}", s) for k, s in shapes])


def _strip(shapes, p):
    return [(k[len(p):], s) for k, s in shapes]


def test_g10_resblock_updown(mg):
    x, emb = torch.from_numpy(mg["in_res_x"]), torch.from_numpy(mg["in_res_emb"])
    for name, cout in (("res_64_128", 128), ("res_64_64", 64)):
        sd_ = _msd(name, _strip(spec._res_shapes("", 64, cout, 256), ""))
        np.testing.assert_allclose(nets._resblock(sd_, f"g10.{name}.", x, emb).numpy(), mg[name], **TOL)
    import torch.nn.functional as F
    sd_ = _msd("up_64", [("conv.weight", (64, 64, 3, 3)), ("conv.bias", (64,))])
    up = nets._conv(sd_, "g10.up_64.conv.", F.interpolate(x, scale_factor=2, mode="nearest"))
    np.testing.assert_allclose(up.numpy(), mg["up_64"], **TOL)
    sd_ = _msd("down_64", [("op.weight", (64, 64, 3, 3)), ("op.bias", (64,))])
    np.testing.assert_allclose(nets._conv(sd_, "g10.down_64.op.", x, stride=2).numpy(), mg["down_64"], **TOL)


def test_g10_attention_modules(mg):
    t, ctx = torch.from_numpy(mg["in_tokens"]), torch.from_numpy(mg["in_ctx"])
    st_shapes = spec._st_shapes("", 128, 96)
    blk = [(k[len("transformer_blocks.0."):], s) for k, s in st_shapes if k.startswith("transformer_blocks.0.")]
    sa = [(k[len("attn1."):], s) for k, s in blk if k.startswith("attn1.")]
    np.testing.assert_allclose(nets._self_attention(_msd("selfattn_128", sa), "g10.selfattn_128.", t, 2).numpy(),
                               mg["selfattn_128"], **TOL)
    xa = [(k[len("t_attn."):], s) for k, s in blk if k.startswith("t_attn.")]
    maps = []
    got = nets._text_cross_attention(_msd("xattn_128", xa), "g10.xattn_128.", t, ctx, 2, "x.", maps)
    np.testing.assert_allclose(got.numpy(), mg["xattn_128"], **TOL)
    np.testing.assert_allclose(maps[0]["attn_map"].numpy(), mg["xattn_128_map"], rtol=1e-4, atol=1e-6)
    single = nets._text_cross_attention(_msd("xattn_128", xa), "g10.xattn_128.", t, ctx[:, :1], 2, "x.", None)
    np.testing.assert_allclose(single.numpy(), mg["xattn_128_single"], **TOL)        # sigmoid branch (L == 1)
    ff = [(k[len("ff."):], s) for k, s in blk if k.startswith("ff.")]
    np.testing.assert_allclose(nets._feed_forward(_msd("ff_128", ff), "g10.ff_128.", t).numpy(), mg["ff_128"], **TOL)
    got = nets._transformer_block(_msd("block_128", blk), "g10.block_128.", t, ctx, 2, "b.", None)
    np.testing.assert_allclose(got.numpy(), mg["block_128"], **TOL)
    xs = torch.from_numpy(mg["in_st_x"])
    got = nets._spatial_transformer(_msd("st_128", st_shapes), "g10.st_128.", xs, ctx, 2, "s.", None)
    np.testing.assert_allclose(got.numpy(), mg["st_128"], **TOL)


def test_g10_vae_modules(mg):
    import torch.nn.functional as F
    x = torch.from_numpy(mg["in_vae_x"])
    for name, cout in (("vres_64_128", 128), ("vres_64_64", 64)):
        sd_ = _msd(name, spec._resnet_shapes("", 64, cout))
        np.testing.assert_allclose(nets._vae_resnet(sd_, f"g10.{name}.", x).numpy(), mg[name], **TOL)
    sd_ = _msd("vattn_64", spec._vae_attn_shapes("", 64))
    np.testing.assert_allclose(nets._vae_attn(sd_, "g10.vattn_64.", x).numpy(), mg["vattn_64"], **TOL)
    sd_ = _msd("vdown_64", [("conv.weight", (64, 64, 3, 3)), ("conv.bias", (64,))])
    got = nets._conv(sd_, "g10.vdown_64.conv.", F.pad(x, (0, 1, 0, 1)), stride=2, padding=0)
    np.testing.assert_allclose(got.numpy(), mg["vdown_64"], **TOL)
    sd_ = _msd("vup_64", [("conv.weight", (64, 64, 3, 3)), ("conv.bias", (64,))])
    got = nets._conv(sd_, "g10.vup_64.conv.", F.interpolate(x, scale_factor=2.0, mode="nearest"))
    np.testing.assert_allclose(got.numpy(), mg["vup_64"], **TOL)


def test_g13_attend_and_excite_gradient(sd, cfg):
    """d local_loss / d x of oracle/backward.py (torch.autograd through the functional UNet) against the gradient
    torch.autograd.grad returned INSIDE the real reference's EulerEDMSampler.attend_and_excite (sampling.py:233-252) — 16x16 latents,
    B = 1, the sampler's step 2 of 10; then the update x - alpha * grad itself."""
    from aae_fixture import aae_batch, aae_functional_weights
    from oracle import backward
    g13 = np.load(os.path.join(GOLD, "aae_golden.npz"))
    batch = aae_batch()                                        # (masks that decide the loss's hard selections: see the fixture)
    torch.manual_seed(1234)
    c, _ = sampling.conditioning(sd, cfg, batch)
    np.testing.assert_allclose(c["concat"].numpy(), g13["g13_c_concat"], **TOL)
    np.testing.assert_allclose(c["t_crossattn"].numpy(), g13["g13_c_txt"], rtol=1e-3, atol=1e-3)
    # The gradient is taken at the GOLDEN's conditioning: it depends on the DIFFERENCES between the label-embedding tokens, which are
    # nearly equal vectors — a conditioning that agrees to 1e-2 (the bf16 label encoder of the HIP path) moves it by 28 %
    # (profiles/r06_aae_debug.txt); the conditioner has its own parity tests (G3 / G6), this one pins the UNet's reverse pass
    c = {"concat": torch.from_numpy(g13["g13_c_concat"]), "t_crossattn": torch.from_numpy(g13["g13_c_txt"])}
    x, sigma = torch.from_numpy(g13["g13_x"]), torch.from_numpy(g13["g13_sigma"])
    assert torch.equal(backward.c_noise_of(sd, sigma), torch.from_numpy(g13["g13_c_noise"]))
    loss, grad = backward.attend_and_excite_grad(sd, cfg, x, sigma, c, batch["mask"], batch["seg_mask"])
    np.testing.assert_allclose(loss.numpy(), g13["g13_local_loss"], rtol=1e-4, atol=1e-6)
    ref = torch.from_numpy(g13["g13_grad"])
    rel = ((grad - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    assert rel < 1e-3, rel                                      # fp32 autograd vs fp32 autograd, different association order
    assert float(ref.abs().max()) > 0.0
    alpha = float(g13["g13_alpha"][0])
    x2 = backward.attend_and_excite(sd, cfg, x, sigma, c, batch["mask"], batch["seg_mask"], alpha, False, 0.0)
    np.testing.assert_allclose(x2.numpy(), g13["g13_x_updated"], rtol=0, atol=2e-6)
    # G13s: the same reverse pass with a dense cotangent on every counted map (smooth functional of the real reference's maps)
    val, gs = backward.maps_functional_grad(sd, cfg, x, sigma, c, aae_functional_weights)
    np.testing.assert_allclose(float(val), float(g13["g13s_value"][0]), rtol=1e-4)
    ref_s = torch.from_numpy(g13["g13s_grad"])
    rel_s = ((gs - ref_s).pow(2).mean().sqrt() / ref_s.pow(2).mean().sqrt()).item()
    assert rel_s < 1e-3, rel_s


def test_g14_training_step_loss_and_parameter_gradients(sd, cfg):
    """SURVEY 8f-4, second half: FullLoss.__call__ (loss.py:131-176) and its gradients with respect to the t_attn / t_norm parameters
    (oracle/training.py, torch.autograd through the functional UNet) against the real reference's loss and autograd gradients at the
    recorded draws of one step (G14), with lambda_local_loss = 0.01 and 0; and G14s: dense cotangents on the counted maps."""
    from aae_fixture import aae_functional_weights, sub, train_batch
    from oracle import training
    g = np.load(os.path.join(GOLD, "train_golden.npz"))
    batch = train_batch()
    z, idx, noise = torch.from_numpy(g["g14_z"]), torch.from_numpy(g["g14_sigma_idx"]), torch.from_numpy(g["g14_noise"])
    cond = {"concat": torch.from_numpy(g["g14_c_concat"]), "t_crossattn": torch.from_numpy(g["g14_c_txt"])}
    names = [str(n) for n in g["g14_names"]]
    assert names == training.trainable_names(sd) and len(names) == 112
    assert sum(sd[n].numel() for n in names) == 75_936_320                      # the 75.9 M trained values of SURVEY 8f-4

    def compare(grads, tag, tol):
        ref_sub = torch.from_numpy(g[f"{tag}_sub"])
        num = den = 0.0
        for i, n in enumerate(names):
            s_ = sub(grads[n])
            num += float((s_ - ref_sub[i, :s_.numel()]).pow(2).sum())
            den += float(ref_sub[i, :s_.numel()].pow(2).sum())
            st = _stats(grads[n])
            np.testing.assert_allclose(st[2], g[f"{tag}_stats"][i, 2], rtol=4 * tol, atol=1e-30)     # sum of squares per tensor
        assert (num / den) ** 0.5 < tol, (tag, (num / den) ** 0.5)

    for tag, lam in (("g14_full", 0.01), ("g14_diff", 0.0)):
        ld, grads = training.training_grads(sd, cfg, z, cond, batch["seg"], batch["seg_mask"], idx, noise, lambda_local=lam)
        if lam > 0:
            for k in ("loss/diff_loss", "loss/local_loss", "loss/full_loss"):
                np.testing.assert_allclose(float(ld[k]), float(g["g14_" + k.replace("/", "_")][0]), rtol=2e-4, atol=1e-7)
        compare(grads, tag, 2e-3)
    val, gs = training.maps_functional_param_grads(sd, cfg, z, cond, idx, noise, aae_functional_weights)
    np.testing.assert_allclose(float(val), float(g["g14s_value"][0]), rtol=1e-4)
    compare(gs, "g14s", 2e-3)
