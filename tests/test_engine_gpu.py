"""End-to-end parity of the MI355X engine (HIP kernels behind the sgm surface) against
  (a) the golden vectors captured from the real reference (tests/golden/*.npz), and
  (b) the fp32 CPU oracle on the same seeded inputs / synthetic weights.

Arithmetic: bf16 storage + MFMA products, fp32 accumulation / statistics / sampler state.  Stated tolerances
(relative to the tensor's RMS, "rel_rms"; and worst element relative to the max magnitude, "rel_max"):
    single network call (UNet eps, VAE, LabelEncoder) : rel_rms <= 2e-2, rel_max <= 8e-2
    10 chaotic Euler steps with random weights (latent): rel_rms <= 6e-2, decoded image <= 4e-2
    the 50-step schedule (benchmarked step count)       : rel_rms <= 3e-2 after 10 / 25 / 50 steps, decoded image <= 4e-2
Measured values are written to gpurun_out/parity_report.txt.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REPORT = os.path.join(ROOT, "gpurun_out", "parity_report.txt")


def _metrics(got, ref):
    got = torch.as_tensor(got).double().cpu()
    ref = torch.as_tensor(ref).double().cpu()
    err = (got - ref)
    rms = ref.pow(2).mean().sqrt().item()
    return err.pow(2).mean().sqrt().item() / max(rms, 1e-30), err.abs().max().item() / max(ref.abs().max().item(), 1e-30)


def _check(name, got, ref, rel_rms, rel_max=None):
    r, m = _metrics(got, ref)
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(f"{name:55s} rel_rms {r:.3e} (tol {rel_rms:.1e})  rel_max {m:.3e}\n")
    assert r <= rel_rms, f"{name}: rel_rms {r:.3e} > {rel_rms}"
    if rel_max is not None:
        assert m <= rel_max, f"{name}: rel_max {m:.3e} > {rel_max}"


# Config #5 (MX8 linears + e4m3 self-attention) — the stated tolerance is DERIVED, not chosen (round 6):
#   * one e4m3 operand with a power-of-two block scale carries 2.7 % relative rms error (3 mantissa bits; tests/test_mx8_gpu.py measures
#     it), an activation x weight product 2.7 % x sqrt(2) = 3.8 %; in a K-term dot product of zero-mean terms the independent errors add
#     like the terms do, so an MX8 GEMM's output carries ~3.8 % whatever K is (test_mx8_linear_vs_torch...: 3.5-3.9e-2 vs unquantised);
#   * what that does to one UNet call is computed on the CPU: tests/error_budget.py injects exactly these roundings into the fp32 oracle
#     (flag L8: the five linears of every 640- / 1280-channel transformer block in their LayerNorm-folded form; A8: the e4m3 attention)
#     on the G7 call -> 3.96e-2 (linears alone), 4.19e-2 with every bf16 rounding of the HIP path and the e4m3 attention on top
#     (profiles/r06_error_budget.txt).
#   Stated: 1.45 x that prediction = 6e-2, for one call AND for the 50-step trajectory (the error must not grow along it).
#   Measured on MI355X: 4.0-4.4e-2 per call, 3.7-3.8e-2 on the trajectory (profiles/r06_parity_report.txt).
TOL_CFG5 = 6e-2


@pytest.fixture(scope="module")
def eg():
    return np.load(os.path.join(GOLD, "engine_golden.npz"))


def _sampler_call(unet, x, ts, tctx, zero_rows):
    """one UNet call EXACTLY as the sampling loop makes it (sampling._Stepper.step -> forward_nhwc): hoisted context k|v folded into
    the fused text cross-attention's tables, the first ``zero_rows`` samples on the zero-context shortcut, no attention maps —
    ``unet(x, t, ctx)`` (the reference signature) takes the map-emitting xattn chain instead.  x fp32 NCHW [B, 9, h, w] -> eps NCHW"""
    from udifftext_amd import ops
    from sgm.modules.diffusionmodules.openaimodel import CPAD
    xin = ops.nchw_to_nhwc(x.float().contiguous(), CPAD)
    emb = unet.time_embedding_rows(ts)
    t_kv = unet.project_context(tctx)
    t_fused = unet.prepare_fused_tattn(t_kv)
    assert all(tb is not None for lst in t_fused for tb in lst)
    eps = unet.forward_nhwc(xin, emb, t_kv, emit_maps=False, zero_ctx_rows=zero_rows, t_fused=t_fused)
    return ops.nhwc_to_nchw(eps, unet.out_channels)


@pytest.fixture(scope="module")
def engine(cuda):
    from udifftext_amd import lib, pipeline
    assert lib.load().udt_device_arch_ok() == 1
    torch.set_grad_enabled(False)
    return pipeline.build_engine(cuda)


@pytest.fixture(scope="module")
def cond256(engine, cuda):
    from udifftext_amd import pipeline, synth
    batch = synth.synthetic_batch(1, 256, 256, 4, seed=0)
    torch.manual_seed(1234)
    batch, buc = pipeline.prepare_batch(batch, cuda)
    c, uc = engine.conditioner.get_unconditional_conditioning(batch, batch_uc=buc, force_uc_zero_embeddings=["label"])
    return batch, c, uc


def test_label_encoder_vs_reference_golden(engine, eg):
    le = engine.conditioner.embedders[0]
    emb = le(["TEXT", "Diffusion", "MI355XNative", "Te9~ é"])
    assert emb.shape == (4, 12, 2048) and emb.dtype == torch.float32
    _check("LabelEncoder (12 layers) vs reference", emb[:, :, ::16].cpu(), eg["g3_label_sub"], 2e-2, 8e-2)


def test_vae_vs_reference_golden(engine, eg, cuda):
    g = torch.Generator().manual_seed(5)
    img = torch.rand((1, 3, 64, 64), generator=g) * 2 - 1
    fs = engine.first_stage_model
    mom = fs.encode_moments(img.to(cuda))
    from udifftext_amd import ops
    _check("VAE encoder moments 64x64 vs reference", ops.nhwc_to_nchw(mom, 8).cpu(), eg["g5_moments"], 2e-2, 8e-2)
    z8 = torch.randn((1, 4, 8, 8), generator=g) * 3.0
    dec = fs.decode(z8.to(cuda))
    _check("VAE decoder 8x8 -> 64x64 vs reference", dec.cpu(), eg["g5_decoded"], 2e-2, 8e-2)


def test_conditioner_vs_reference_golden(cond256, eg):
    _, c, uc = cond256
    assert c["concat"].shape == (1, 5, 32, 32) and uc["t_crossattn"].abs().max().item() == 0.0
    _check("conditioner c.concat vs reference", c["concat"].cpu(), eg["g6_c_concat"], 2e-2, 8e-2)
    _check("conditioner uc.concat vs reference", uc["concat"].cpu(), eg["g6_uc_concat"], 2e-2, 8e-2)
    _check("conditioner c.t_crossattn vs reference", c["t_crossattn"][:, :, ::16].cpu(), eg["g6_c_txt_sub"], 2e-2, 8e-2)
    np.testing.assert_allclose(c["concat"][:, :1].cpu().numpy(), eg["g6_c_concat"][:, :1], atol=1e-6)   # mask: exact
    assert not torch.allclose(c["concat"][:, 1:], uc["concat"][:, 1:])      # different posterior noise (RNG contract)


def test_unet_call_vs_reference_golden(engine, cond256, eg, cuda):
    batch, _, _ = cond256
    # feed the REFERENCE's conditioning so that only the UNet is under test
    x7 = torch.from_numpy(eg["g7_x"]).to(cuda)
    ucc, cc = torch.from_numpy(eg["g6_uc_concat"]).to(cuda), torch.from_numpy(eg["g6_c_concat"]).to(cuda)
    le = engine.conditioner.embedders[0]
    tctx = torch.cat([torch.zeros((1, 12, 2048), device=cuda), le(batch["label"])])
    xin = torch.cat([torch.cat([x7, x7]), torch.cat([ucc, cc])], dim=1)
    unet = engine.model.diffusion_model
    eps = unet(xin, timesteps=torch.tensor([999, 999], device=cuda), t_context=tctx)
    assert eps.shape == (2, 4, 32, 32) and eps.dtype == torch.float32
    _check("UNet eps (CFG pair, 32x32 latent) vs reference", eps.cpu(), eg["g7_eps"], 2e-2, 8e-2)
    names = json.load(open(os.path.join(GOLD, "attn_map_names.json")))
    for item, (name, heads, size, shape) in zip(unet.attn_map_cache, names):
        assert item["name"] == name and item["size"] == size and list(item["attn_map"].shape) == shape
        f = item["attn_map"].float().reshape(-1)
        step = max(1, f.numel() // 2048)
        _check(f"t_attn map {name}", f[::step][:2048].cpu(), eg[f"g7_attn_{name}_sub"], 3e-2)
    ll = engine.loss_fn.get_min_local_loss(unet.attn_map_cache, batch["mask"], batch["seg_mask"])
    _check("get_min_local_loss vs reference", ll.cpu(), eg["g8_local_loss"], 3e-2)


def test_unet_and_vae_with_fused_groupnorm_vs_reference_golden(engine, cond256, eg, cuda):
    """the opt-in fused chain (UDT_FUSE_GN=1: statistics from producer epilogues -> udt_gn_finalize -> GroupNorm + SiLU on
    the convolution's staged patch, udt_gn_silu_conv3x3_fwd) through the whole UNet and the VAE, against the same
    reference goldens and tolerances as the default path"""
    import sgm.modules.hipnn as H
    from udifftext_amd import ops
    batch, _, _ = cond256
    prev = H.FUSE_GN
    H.FUSE_GN = True
    try:
        x7 = torch.from_numpy(eg["g7_x"]).to(cuda)
        ucc, cc = torch.from_numpy(eg["g6_uc_concat"]).to(cuda), torch.from_numpy(eg["g6_c_concat"]).to(cuda)
        le = engine.conditioner.embedders[0]
        tctx = torch.cat([torch.zeros((1, 12, 2048), device=cuda), le(batch["label"])])
        xin = torch.cat([torch.cat([x7, x7]), torch.cat([ucc, cc])], dim=1)
        ops.WORK_COUNTER = {}
        eps = engine.model.diffusion_model(xin, timesteps=torch.tensor([999, 999], device=cuda), t_context=tctx)
        n_fused = ops.WORK_COUNTER.get("fused_gn_convs", 0)
        ops.WORK_COUNTER = None
        assert n_fused >= 30, f"only {n_fused} convolutions took the fused GroupNorm path"
        _check("UNet eps with fused GroupNorm vs reference", eps.cpu(), eg["g7_eps"], 2e-2, 8e-2)
        g = torch.Generator().manual_seed(5)
        img = torch.rand((1, 3, 64, 64), generator=g) * 2 - 1
        fs = engine.first_stage_model
        _check("VAE encoder moments with fused GroupNorm vs reference", ops.nhwc_to_nchw(fs.encode_moments(img.to(cuda)), 8).cpu(),
               eg["g5_moments"], 2e-2, 8e-2)
        z8 = torch.randn((1, 4, 8, 8), generator=g) * 3.0
        _check("VAE decoder with fused GroupNorm vs reference", fs.decode(z8.to(cuda)).cpu(), eg["g5_decoded"], 2e-2, 8e-2)
    finally:
        H.FUSE_GN = prev


def test_unet_call_fp8_linears_vs_reference_golden(engine, cond256, eg, cuda):
    """BASELINE config #5's arithmetic (UDT_FP8=1, second generation): every linear of the 640- / 1280-channel transformer blocks
    on MX8 operands (e4m3 weights with per-channel scales, e4m3 activations with E8M0 block scales written by the producers'
    epilogues, fp32 accumulation), through the call path of the sampling loop (fused text cross-attention, zero-context shortcut
    for the unconditional sample).  Stated tolerance against the fp32 reference golden: rel_rms <= TOL_CFG5 = 6e-2 (derived above; e4m3 keeps 3 mantissa
    bits: each quantised product carries ~3.6 % error, tests/test_mx8_gpu.py; bf16 path: 2e-2 stated / 1.4e-2 measured); the
    eps must also stay within that of the bf16 path.  The e4m3 attention is switched OFF here (hipnn.FP8_ATTENTION = False, asserted through the launch counter): this is the linears-only leg."""
    import sgm.modules.hipnn as H
    from udifftext_amd import ops
    batch, _, _ = cond256
    x7 = torch.from_numpy(eg["g7_x"]).to(cuda)
    ucc, cc = torch.from_numpy(eg["g6_uc_concat"]).to(cuda), torch.from_numpy(eg["g6_c_concat"]).to(cuda)
    le = engine.conditioner.embedders[0]
    tctx = torch.cat([torch.zeros((1, 12, 2048), device=cuda), le(batch["label"])])
    xin = torch.cat([torch.cat([x7, x7]), torch.cat([ucc, cc])], dim=1)
    ts = torch.tensor([999, 999], device=cuda)
    unet = engine.model.diffusion_model
    ref_bf16 = _sampler_call(unet, xin, ts, tctx, 1)
    _check("UNet eps on the sampling loop's call path (bf16) vs reference", ref_bf16.cpu(), eg["g7_eps"], 2e-2, 8e-2)
    prev, prev_a8 = H.FP8_LINEARS, H.FP8_ATTENTION
    H.FP8_LINEARS, H.FP8_ATTENTION = True, False
    try:
        ops.WORK_COUNTER = {}
        eps = _sampler_call(unet, xin, ts, tctx, 1)
        n8 = ops.WORK_COUNTER.get("gemm_fp8_launches", 0)
        assert ops.WORK_COUNTER.get("attn_fp8_launches", 0) == 0 and ops.WORK_COUNTER.get("attn_launches", 0) == 16
        ops.WORK_COUNTER = {}
        eps_ref_sig = unet(xin, timesteps=ts, t_context=tctx)        # the reference signature: map-emitting xattn chain, bf16 feed-forward
        n8_ref_sig = ops.WORK_COUNTER.get("gemm_fp8_launches", 0)
        assert ops.WORK_COUNTER.get("attn_fp8_launches", 0) == 0
        H.FP8_ATTENTION = True                                       # the complete config #5: + attn1 on e4m3 operands
        ops.WORK_COUNTER = {}
        eps_a8 = _sampler_call(unet, xin, ts, tctx, 1)
        assert ops.WORK_COUNTER.get("attn_fp8_launches", 0) == 16 and ops.WORK_COUNTER.get("attn_launches", 0) == 0
        ops.WORK_COUNTER = None
    finally:
        H.FP8_LINEARS, H.FP8_ATTENTION = prev, prev_a8
    # 10 transformer blocks of width 640 / 1280 x (q|k|v, to_out, GEGLU, ff.net[2], proj_out) + the middle block, whose 4 x 4 map
    # (16 tokens per sample) is below the fused text cross-attention's token tile: unfused t_attn, bf16 feed-forward and proj_out
    assert n8 == 10 * 5 + 2, f"{n8} MX8 GEMM launches on the sampling loop's call path"
    assert n8_ref_sig == 11 * 2, f"{n8_ref_sig} MX8 GEMM launches through the reference signature (q|k|v, to_out)"
    _check("UNet eps with MX8 linears ONLY (bf16 attention) vs reference", eps.cpu(), eg["g7_eps"], TOL_CFG5)
    _check("UNet eps with MX8 linears ONLY vs the bf16 path", eps.cpu(), ref_bf16.cpu(), TOL_CFG5)
    _check("UNet eps with MX8 linears ONLY through unet(x, t, ctx) vs reference", eps_ref_sig.cpu(), eg["g7_eps"], TOL_CFG5)
    _check("UNet eps with MX8 linears + e4m3 attention (config #5) vs reference", eps_a8.cpu(), eg["g7_eps"], TOL_CFG5)
    assert not torch.equal(eps_a8, eps), "the e4m3 attention leg ran the bf16 attention"


def test_zero_context_shortcut_is_bit_exact(engine, cond256, cuda):
    """the sampler skips the t_attn GEMMs of the unconditional half (context == 0 -> x + to_out.bias); the eps must
    be bit-identical to running the full cross-attention on the zero context"""
    from udifftext_amd import ops, packing
    _, c, uc = cond256
    assert not bool(uc["t_crossattn"].any()), "force_uc_zero_embeddings must zero the label context"
    unet = engine.model.diffusion_model
    torch.manual_seed(7)
    xin = torch.zeros((2, 32, 32, packing.KPAD), dtype=torch.bfloat16, device=cuda)
    xin[..., :9] = torch.randn((2, 32, 32, 9), device=cuda).bfloat16()
    emb = unet.time_embedding_rows(torch.tensor([500.0, 500.0], device=cuda))
    t_kv = unet.project_context(torch.cat((uc["t_crossattn"], c["t_crossattn"]), 0))
    full = unet.forward_nhwc(xin, emb, t_kv, zero_ctx_rows=0)
    fast = unet.forward_nhwc(xin, emb, t_kv, zero_ctx_rows=1)
    assert torch.equal(full, fast)
    # and the bias_add kernel on its own
    x = torch.randn((64, 320), device=cuda).bfloat16()
    b = torch.randn((320,), device=cuda)
    assert torch.equal(ops.bias_add(x, b), (x.float() + b).bfloat16())


def test_per_block_activations_vs_oracle(engine, cond256, eg, cuda):
    """per-block parity inside the UNet (taps of the oracle == reference goldens, see test_oracle_golden)"""
    from udifftext_amd import ops
    unet = engine.model.diffusion_model
    taps = {}
    hooks = []
    for i, blk in enumerate(unet.input_blocks):
        hooks.append(blk.register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(f"input_blocks.{i}", o)))
    hooks.append(unet.middle_block.register_forward_hook(lambda m, a, o: taps.__setitem__("middle_block", o)))
    for i, blk in enumerate(unet.output_blocks):
        hooks.append(blk.register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(f"output_blocks.{i}", o)))
    batch, c, uc = cond256
    x7 = torch.from_numpy(eg["g7_x"]).to(cuda)
    xin = torch.cat([torch.cat([x7, x7]), torch.cat([uc["concat"], c["concat"]])], dim=1)
    tctx = torch.cat([uc["t_crossattn"], c["t_crossattn"]])
    # hooks need module __call__: run the blocks through forward_nhwc's module calls
    unet(xin, timesteps=torch.tensor([999, 999], device=cuda), t_context=tctx)
    for h in hooks:
        h.remove()
    if not taps:
        pytest.skip("block hooks not triggered by forward_nhwc")
    for k, v in taps.items():
        nchw = ops.nhwc_to_nchw(v.contiguous(), v.shape[-1])
        f = nchw.float().reshape(-1)
        step = max(1, f.numel() // 2048)
        _check(f"UNet block {k}", f[::step][:2048].cpu(), eg[f"g7_tap_{k}_sub"], 3e-2)


def test_ten_step_sampling_vs_reference_golden(engine, cond256, eg, cuda):
    """BASELINE config #1: 256x256, 10 steps, 'TEXT', batch 1, CFG 5 — latent after the full loop + decode"""
    from udifftext_amd import config as C, pipeline
    batch, c, uc = cond256
    sampler = pipeline.init_sampling(10, 5.0, cuda)
    cfgs = C.default_runtime_config(steps=10, batch_size=1, noise_iters=0)
    torch.manual_seed(99)
    x0 = sampler.get_init_noise(cfgs, engine, cond=c, batch=batch, uc=uc)
    np.testing.assert_array_equal(x0.cpu().numpy(), eg["g9_x0"])          # CPU RNG draw order
    z = sampler(engine, x0.clone(), cond=c, batch=batch, uc=uc)
    _check("10-step latent (config #1) vs reference", z.cpu(), eg["g9_latent"], 6e-2)
    dec = engine.decode_first_stage(z)
    _check("decoded image of the 10-step latent vs reference", dec[:, :, ::8, ::8].cpu(), eg["g9_decoded_sub"], 4e-2)


def test_fifty_step_sampling_vs_reference_golden(engine, cond256, cuda):
    """the benchmarked step count (configs/test.yaml:20): 50 Euler steps, 256x256, "TEXT", batch 1, CFG 5 against the
    trajectory of the REAL reference (tests/golden/engine_golden_50.npz, make_golden.py --g11).  With random weights the
    stated tolerance is rel_rms <= 3e-2 at every horizon (10 / 25 / 50 steps; measured on MI355X: 1.0e-2 at all three —
    the error does not grow along the trajectory) and <= 4e-2 for the decoded image (measured 1.8e-2)."""
    from udifftext_amd import config as C, pipeline
    g11 = np.load(os.path.join(GOLD, "engine_golden_50.npz"))
    batch, c, uc = cond256
    cfgs = C.default_runtime_config(steps=50, batch_size=1, noise_iters=0)
    for horizon, tol in ((10, 3e-2), (25, 3e-2), (50, 3e-2)):
        sampler = pipeline.init_sampling(50, 5.0, cuda)
        torch.manual_seed(4242)
        x0 = sampler.get_init_noise(cfgs, engine, cond=c, batch=batch, uc=uc)
        np.testing.assert_array_equal(x0.cpu().numpy(), g11["g11_x0"])
        if horizon == 50:
            z = sampler(engine, x0.clone(), cond=c, batch=batch, uc=uc)
        else:                                   # the first `horizon` steps of the 50-step schedule, eager launches
            from sgm.modules.diffusionmodules.sampling import _Stepper
            sig = sampler._host_sigmas()
            z = x0.clone().float() * (1.0 + sig[0] ** 2.0) ** 0.5
            st = _Stepper(engine, c, uc, 1, z.shape[2:], 5.0)
            for i in range(horizon):
                st.step(z, sig[i], sig[i + 1])
            st.check()
        _check(f"50-step schedule, latent after {horizon} steps vs reference", z.cpu(), g11[f"g11_latent_{horizon}"], tol)
    dec = engine.decode_first_stage(z)
    _check("decoded image of the 50-step latent vs reference", dec[:, :, ::8, ::8].cpu(), g11["g11_decoded_sub"], 4e-2)


def test_config2_512_fifty_steps_vs_reference_golden(engine, cuda):
    """BASELINE config #2 end to end on one image: 512x512, the 9-character label, 50 Euler steps, CFG 5 — conditioning,
    x0 (CPU RNG contract, bit-exact) and the latent after 10 / 25 / 50 steps against the trajectory of the REAL reference's
    EulerEDMSampler.__call__ (tests/golden/engine_golden_512.npz, make_golden.py --g12), then the decoded image.
    Stated tolerances as for the 256x256 trajectory: rel_rms <= 3e-2 at every horizon, decoded image <= 4e-2."""
    from udifftext_amd import config as C, pipeline, synth
    g12 = np.load(os.path.join(GOLD, "engine_golden_512.npz"))
    batch = synth.synthetic_batch(1, 512, 512, 9, seed=12)
    torch.manual_seed(1234)
    batch, buc = pipeline.prepare_batch(batch, cuda)
    c, uc = engine.conditioner.get_unconditional_conditioning(batch, batch_uc=buc, force_uc_zero_embeddings=["label"])
    _check("512x512 conditioner c.concat vs reference", c["concat"].cpu(), g12["g12_c_concat"], 2e-2, 8e-2)
    _check("512x512 conditioner uc.concat vs reference", uc["concat"].cpu(), g12["g12_uc_concat"], 2e-2, 8e-2)
    _check("512x512 conditioner c.t_crossattn vs reference", c["t_crossattn"][:, :, ::16].cpu(), g12["g12_c_txt_sub"], 2e-2, 8e-2)
    cfgs = C.default_runtime_config(steps=50, batch_size=1, noise_iters=0)
    for horizon in (10, 25, 50):
        sampler = pipeline.init_sampling(50, 5.0, cuda)
        torch.manual_seed(512)
        x0 = sampler.get_init_noise(cfgs, engine, cond=c, batch=batch, uc=uc)
        np.testing.assert_array_equal(x0.cpu().numpy(), g12["g12_x0"])
        if horizon == 50:
            z = sampler(engine, x0.clone(), cond=c, batch=batch, uc=uc)
        else:
            from sgm.modules.diffusionmodules.sampling import _Stepper
            sig = sampler._host_sigmas()
            z = x0.clone().float() * (1.0 + sig[0] ** 2.0) ** 0.5
            st = _Stepper(engine, c, uc, 1, z.shape[2:], 5.0)
            for i in range(horizon):
                st.step(z, sig[i], sig[i + 1])
            st.check()
        _check(f"512x512, 50-step schedule, latent after {horizon} steps vs reference", z.cpu(), g12[f"g12_latent_{horizon}"], 3e-2)
    dec = engine.decode_first_stage(z)
    _check("512x512 decoded image of the 50-step latent vs reference", dec[:, :, ::8, ::8].cpu(), g12["g12_decoded_sub"], 4e-2)


def test_unet_call_at_benchmarked_shape_vs_oracle(engine, cuda):
    """BASELINE config #2's UNet call: 64x64 latents, batch 4 -> 8 samples (uc half first, zero text context), the
    tile / stream-K plans of the benchmarked shape — against the fp32 CPU oracle on the same weights.  The oracle
    evaluates 2 + 2 of the 8 samples (samples are independent: per-sample GroupNorm / LayerNorm / attention)."""
    from oracle import nets, spec
    from udifftext_amd import synth
    torch.manual_seed(31)
    B = 4
    le = engine.conditioner.embedders[0]
    ctx = le(synth.synthetic_batch(B, 512, 512, 9, seed=6)["label"])
    tctx = torch.cat([torch.zeros_like(ctx), ctx])
    x = torch.randn((2 * B, 9, 64, 64), device=cuda)
    ts = torch.full((2 * B,), 441, device=cuda)
    eps = engine.model.diffusion_model(x, timesteps=ts, t_context=tctx)
    assert eps.shape == (2 * B, 4, 64, 64)
    pick = [0, 3, 4, 7]
    sd = {k: v.detach().float().cpu() for k, v in engine.state_dict().items() if k.startswith("model.")}
    with torch.no_grad():
        ref = nets.unet_forward(sd, x[pick].cpu(), ts[pick].cpu(), tctx[pick].float().cpu(), spec.EngineConfig().unet)
    _check("UNet eps at 64x64 latents, 8 samples (config #2 call) vs oracle", eps[pick].cpu(), ref, 2e-2, 8e-2)


def test_unet_call_in_flight_shape_vs_oracle(engine, cuda):
    """the throughput-mode call: 32 samples per UNet call (4 batches of 4 concatenated), planned for half of the CUs
    (cu_share 2) on a side stream with its own workspace WHILE a second stream runs another call — exactly how
    EulerEDMSampler.sample_in_flight launches it — against the oracle on 4 of the 32 samples"""
    from oracle import nets, spec
    from udifftext_amd import ops, packing, synth
    torch.manual_seed(32)
    unet = engine.model.diffusion_model
    n = 16
    le = engine.conditioner.embedders[0]
    labels = [synth.synthetic_label(9, i) for i in range(n)]
    ctx = le(labels)
    tctx = torch.cat([torch.zeros_like(ctx), ctx])
    x = torch.randn((2 * n, 9, 64, 64), device=cuda)
    ts = torch.full((2 * n,), 701.0, device=cuda)
    xin = ops.nchw_to_nhwc(x.float().contiguous(), packing.KPAD)
    emb = unet.time_embedding_rows(ts)
    t_kv = unet.project_context(tctx)
    s1, s2 = torch.cuda.Stream(device=cuda), torch.cuda.Stream(device=cuda)
    w1, w2 = ops.Workspace(cuda), ops.Workspace(cuda)
    torch.cuda.synchronize()
    with torch.cuda.stream(s2), ops.launch_context(cu_share=2, workspace=w2):
        other = unet.forward_nhwc(xin.flip(0).contiguous(), emb, t_kv, zero_ctx_rows=0)
    with torch.cuda.stream(s1), ops.launch_context(cu_share=2, workspace=w1):
        eps = unet.forward_nhwc(xin, emb, t_kv, zero_ctx_rows=n)
    torch.cuda.synchronize()
    w1.check(); w2.check()
    assert torch.isfinite(other).all()
    got = ops.nhwc_to_nchw(eps, 4)
    pick = [0, 15, 16, 31]
    sd = {k: v.detach().float().cpu() for k, v in engine.state_dict().items() if k.startswith("model.")}
    with torch.no_grad():
        ref = nets.unet_forward(sd, x[pick].cpu(), ts[pick].long().cpu(), tctx[pick].float().cpu(), spec.EngineConfig().unet)
    _check("UNet eps, 32 samples per call under cu_share 2 on two streams vs oracle", got[pick].cpu(), ref, 2e-2, 8e-2)


def test_vae_at_512_vs_oracle(engine, cuda):
    """the benchmarked VAE shapes: encode of one 512x512 image (moments) and decode of one 64x64 latent (N = 4096
    single-head attention, 512x512x128 convolutions) against the oracle"""
    from oracle import nets, spec
    from udifftext_amd import ops, synth
    cfg = spec.EngineConfig()
    batch = synth.synthetic_batch(1, 512, 512, 9, seed=8)
    fs = engine.first_stage_model
    mom = ops.nhwc_to_nchw(fs.encode_moments(batch["masked"].to(cuda)), 8)
    sd = {k: v.detach().float().cpu() for k, v in engine.state_dict().items() if k.startswith("first_stage_model.")}
    with torch.no_grad():
        ref_m = nets.vae_encode_moments(sd, batch["masked"], cfg.vae)
    _check("VAE encoder moments at 512x512 vs oracle", mom.cpu(), ref_m, 2e-2, 8e-2)
    g = torch.Generator().manual_seed(9)
    z = torch.randn((1, 4, 64, 64), generator=g) * 3.0
    dec = fs.decode(z.to(cuda))
    with torch.no_grad():
        ref_d = nets.vae_decode(sd, z, cfg.vae)
    assert dec.shape == (1, 3, 512, 512)
    _check("VAE decoder 64x64 -> 512x512 vs oracle", dec.cpu(), ref_d, 2e-2, 8e-2)


def test_vae_decode_flash_attention_d512_vs_oracle_and_block_form(engine, cuda, monkeypatch):
    """the VAE mid-block attention through the flash kernel (a single 64x64 latent: the key-split form, udt_attn512_split_fwd)
    against the oracle, and against the GEMM -> softmax -> GEMM block form on the same latent"""
    from oracle import nets, spec
    from sgm.modules.diffusionmodules import model as vmodel
    cfg = spec.EngineConfig()
    fs = engine.first_stage_model
    sd = {k: v.detach().float().cpu() for k, v in engine.state_dict().items() if k.startswith("first_stage_model.")}
    g = torch.Generator().manual_seed(19)
    z = torch.randn((1, 4, 64, 64), generator=g) * 3.0
    with torch.no_grad():
        ref_d = nets.vae_decode(sd, z, cfg.vae)
    monkeypatch.setattr(vmodel, "ATTN_FLASH_512", "1")
    assert vmodel._flash512(4, 4096) and vmodel._flash512(1, 4096)
    dec_flash = fs.decode(z.to(cuda))
    monkeypatch.setattr(vmodel, "ATTN_FLASH_512", "0")
    dec_block = fs.decode(z.to(cuda))
    _check("VAE decoder 64x64 -> 512x512, flash attention (head_dim 512) vs oracle", dec_flash.cpu(), ref_d, 2e-2, 8e-2)
    _check("VAE decoder: flash attention vs block form", dec_flash.cpu(), dec_block.cpu(), 1e-2, 4e-2)


def test_graph_replay_matches_eager_launches(engine, cond256, cuda):
    """the sampler's hipGraph path (capture once per step index, replay; static conditioning buffers refreshed by
    rebind for the next batch) must give exactly the eager launch sequence's latent"""
    from udifftext_amd import pipeline, synth
    batch, c, uc = cond256
    torch.manual_seed(5)
    x0 = torch.randn((1, 4, 32, 32), device=cuda)
    eager = pipeline.init_sampling(4, 5.0, cuda)
    eager.use_graphs = False
    graphed = pipeline.init_sampling(4, 5.0, cuda)
    graphed.use_graphs = True
    ze = eager(engine, x0.clone(), cond=c, batch=batch, uc=uc)
    zg = graphed(engine, x0.clone(), cond=c, batch=batch, uc=uc)
    assert graphed.use_graphs and len(graphed._graphed) == 1, "graph capture fell back to eager launches"
    assert torch.equal(ze, zg)
    # a second batch with other conditioning re-uses the captured graphs through rebind()
    b2 = synth.synthetic_batch(1, 256, 256, 4, seed=3)
    b2, buc2 = pipeline.prepare_batch(b2, cuda)
    c2, uc2 = engine.conditioner.get_unconditional_conditioning(b2, batch_uc=buc2, force_uc_zero_embeddings=["label"])
    gs = next(iter(graphed._graphed.values()))
    n_graphs = len(gs.graphs)
    ze2 = eager(engine, x0.clone(), cond=c2, batch=b2, uc=uc2)
    zg2 = graphed(engine, x0.clone(), cond=c2, batch=b2, uc=uc2)
    assert next(iter(graphed._graphed.values())) is gs and len(gs.graphs) == n_graphs
    assert torch.equal(ze2, zg2)
    assert not torch.equal(ze, ze2)


def test_batches_in_flight_match_sequential(engine, cond256, cuda):
    """two batches sampled concurrently (two launch streams, each planned for half the CUs) against the same two
    batches one after the other; the stream-K split points differ, so agreement is to fp32-accumulation order"""
    from udifftext_amd import pipeline, synth
    _, c, uc = cond256
    b2 = synth.synthetic_batch(1, 256, 256, 4, seed=3)
    b2, buc2 = pipeline.prepare_batch(b2, cuda)
    c2, uc2 = engine.conditioner.get_unconditional_conditioning(b2, batch_uc=buc2, force_uc_zero_embeddings=["label"])
    torch.manual_seed(11)
    xa, xb = torch.randn((1, 4, 32, 32), device=cuda), torch.randn((1, 4, 32, 32), device=cuda)
    seq = pipeline.init_sampling(4, 5.0, cuda)
    seq.use_graphs = False
    za, zb = seq(engine, xa.clone(), cond=c, uc=uc), seq(engine, xb.clone(), cond=c2, uc=uc2)
    par = pipeline.init_sampling(4, 5.0, cuda)
    for _ in range(2):                                  # second round replays the captured graphs through rebind()
        ya, yb = par.sample_in_flight(engine, [xa.clone(), xb.clone()], [c, c2], [uc, uc2])
        _check("2 batches in flight, batch A vs sequential", ya.cpu(), za.cpu(), 2e-2)
        _check("2 batches in flight, batch B vs sequential", yb.cpu(), zb.cpu(), 2e-2)
    assert len(par._in_flight) == 2 and par.use_graphs


def test_noise_search_vs_reference_golden(engine, cond256, eg, cuda):
    from udifftext_amd import config as C, pipeline
    batch, c, uc = cond256
    sampler = pipeline.init_sampling(10, 5.0, cuda)
    cfgs = C.default_runtime_config(steps=10, batch_size=1, noise_iters=2)
    torch.manual_seed(77)
    xs = sampler.get_init_noise(cfgs, engine, cond=c, batch=batch, uc=uc)
    np.testing.assert_array_equal(xs.cpu().numpy(), eg["g9_search_x0"])    # same candidate wins


def test_full_size_properties(engine, cuda):
    """512x512 (BASELINE config #2 shapes, batch 2): size-independent properties of the path"""
    from udifftext_amd import config as C, pipeline, synth
    B = 2
    sampler = pipeline.init_sampling(3, 5.0, cuda)
    cfgs = C.default_runtime_config(steps=3, batch_size=B, noise_iters=0)
    torch.manual_seed(3)
    s1, z1 = pipeline.predict(cfgs, engine, sampler, synth.synthetic_batch(B, 512, 512, 9, seed=1))
    torch.manual_seed(3)
    s2, z2 = pipeline.predict(cfgs, engine, sampler, synth.synthetic_batch(B, 512, 512, 9, seed=1))
    assert s1.shape == (B, 3, 512, 512) and z1.shape == (B, 4, 64, 64)
    assert torch.isfinite(s1).all() and float(s1.min()) >= 0.0 and float(s1.max()) <= 1.0
    assert torch.equal(z1, z2) and torch.equal(s1, s2)                      # deterministic given the seed
    # batch independence: image 0 alone gives the same latent as image 0 inside the batch of 2
    torch.manual_seed(3)
    b1 = synth.synthetic_batch(B, 512, 512, 9, seed=1)
    one = {k: (v[:1] if isinstance(v, torch.Tensor) else v[:1]) for k, v in b1.items()}
    cfg1 = C.default_runtime_config(steps=3, batch_size=1, noise_iters=0)
    # same CPU draws for sample 0: c-noise, uc-noise, x0 are drawn per batch, so draw with B and slice
    torch.manual_seed(3)
    nc, nuc, nx = torch.randn(B, 4, 64, 64), torch.randn(B, 4, 64, 64), torch.randn(B, 4, 64, 64)
    import unittest.mock as mock
    draws = iter([nc[:1], nuc[:1], nx[:1]])
    # (the noise source draws into pinned host memory: torch.randn(shape, out=...))
    with mock.patch("torch.randn", side_effect=lambda *a, **k: (k["out"].copy_(next(draws)) if k.get("out") is not None else next(draws))):
        _, z_one = pipeline.predict(cfg1, engine, sampler, one)
    r, _ = _metrics(z_one.cpu(), z1[:1].cpu())
    # different batch => different tile / split-K plans => bf16-rounding-level differences, amplified over 3 steps
    assert r < 5e-2, f"batch dependence: {r}"


def test_unet_call_at_768_vs_oracle(engine, cuda):
    """BASELINE config #4 geometry (768x768 -> 96x96 latents, 12-character context): one UNet call on a CFG pair
    against the fp32 CPU oracle with the same synthetic weights.  96 is not a power of two: other convolution
    tilings (3 patch tiles per row), 9216-token self-attention, ragged stream-K shares."""
    from oracle import nets, spec
    from udifftext_amd import synth
    torch.manual_seed(21)
    le = engine.conditioner.embedders[0]
    batch = synth.synthetic_batch(1, 768, 768, 12, seed=5)
    ctx = le(batch["label"])
    tctx = torch.cat([torch.zeros_like(ctx), ctx])
    x = torch.randn((2, 9, 96, 96), device=cuda)
    ts = torch.tensor([601, 601], device=cuda)
    eps = engine.model.diffusion_model(x, timesteps=ts, t_context=tctx)
    assert eps.shape == (2, 4, 96, 96)
    sd = {k: v.detach().float().cpu() for k, v in engine.state_dict().items()}
    with torch.no_grad():
        ref = nets.unet_forward(sd, x.cpu(), ts.cpu(), tctx.float().cpu(), spec.EngineConfig().unet)
    _check("UNet eps at 96x96 latents (config #4) vs oracle", eps.cpu(), ref, 2e-2, 8e-2)


def test_768_path_properties(engine, cuda):
    """config #4 end to end (768x768, 12 characters, batch 2, 2 steps): shapes, range, determinism"""
    from udifftext_amd import config as C, pipeline, synth
    sampler = pipeline.init_sampling(2, 5.0, cuda)
    cfgs = C.default_runtime_config(steps=2, batch_size=2, noise_iters=0)
    outs = []
    for _ in range(2):
        torch.manual_seed(4)
        outs.append(pipeline.predict(cfgs, engine, sampler, synth.synthetic_batch(2, 768, 768, 12, seed=2)))
    (s1, z1), (s2, z2) = outs
    assert s1.shape == (2, 3, 768, 768) and z1.shape == (2, 4, 96, 96)
    assert torch.isfinite(s1).all() and float(s1.min()) >= 0.0 and float(s1.max()) <= 1.0
    assert torch.equal(z1, z2) and torch.equal(s1, s2)


def test_predict_many_matches_predict(engine, cuda):
    """throughput mode (2 batches fused per sampling batch x 2 launch streams) against predict() batch by batch:
    same CPU noise draws in the same order, per-image results equal up to the launch plans' summation order; also a
    ragged tail (5 batches -> group of 4 + single)"""
    from udifftext_amd import config as C, pipeline, synth
    cfgs = C.default_runtime_config(steps=3, batch_size=1, noise_iters=0)
    batches = [synth.synthetic_batch(1, 256, 256, 4, seed=40 + i) for i in range(5)]
    seq = pipeline.init_sampling(3, 5.0, cuda)
    torch.manual_seed(8)
    ref = [pipeline.predict(cfgs, engine, seq, b) for b in batches]
    par = pipeline.init_sampling(3, 5.0, cuda)
    torch.manual_seed(8)
    got = pipeline.predict_many(cfgs, engine, par, batches, in_flight=2, fuse=2)
    assert len(got) == len(ref)
    for i, ((s_ref, z_ref), (s_got, z_got)) in enumerate(zip(ref, got)):
        assert s_got.shape == s_ref.shape and z_got.shape == z_ref.shape
        _check(f"predict_many latent of batch {i} vs predict", z_got.cpu(), z_ref.cpu(), 3e-2)
        _check(f"predict_many image of batch {i} vs predict", s_got.cpu(), s_ref.cpu(), 3e-2)


@pytest.mark.parametrize("fp8", [False, True])
def test_sample_alone_vs_inside_a_batch(engine, cuda, monkeypatch, fp8):
    """What a sample's result owes to the batch it is computed in (bf16, and config #5: MX8 linears + e4m3 self-attention).  No value of
    another sample enters it — GroupNorm / LayerNorm statistics, MX block scales and attention are per sample or per row, the e4m3 v
    multiplier is per layer and data-free — and the same call twice is bit-identical.  But the batch size selects the launch PLANS:
    8 x 4096 tokens run the row-resident LayerNorm-folded projections (their own GEGLU / fold arithmetic), 2 x 4096 the tiled
    kernels; each is within the stated tolerance of the reference, and they differ from each other by a few 1e-4 per block, adding up
    over the 35 blocks of a call to about the arithmetic's own distance from fp32 (tools/batch_dependence.py,
    profiles/r05_batch_dependence.txt: the first convolution and ResBlock are bit-identical, the first transformer block brings 6e-4).
    Stated: one call, a sample alone vs inside a batch of 4 (and of 2): bf16 2e-2 (measured 1.1e-2), config #5 6e-2 (3.1e-2)."""
    import sgm.modules.hipnn as H
    from udifftext_amd import synth
    monkeypatch.setattr(H, "FP8_LINEARS", fp8)
    monkeypatch.setattr(H, "FP8_ATTENTION", fp8)
    torch.manual_seed(35)
    B = 4
    le = engine.conditioner.embedders[0]
    ctx = le(synth.synthetic_batch(B, 512, 512, 9, seed=16)["label"])
    x = torch.randn((B, 9, 64, 64), device=cuda)
    unet = engine.model.diffusion_model

    def call(idx):
        n = len(idx)
        xs = torch.cat([x[idx], x[idx]])
        tc = torch.cat([torch.zeros_like(ctx[idx]), ctx[idx]])
        return _sampler_call(unet, xs, torch.full((2 * n,), 441, device=cuda), tc, n)
    tol = 6e-2 if fp8 else 2e-2
    tag = "config #5" if fp8 else "bf16"
    eps4, eps1 = call([0, 1, 2, 3]), call([2])
    _check(f"{tag}: one sample alone vs inside a batch of 4 (one UNet call)", eps1.cpu(), eps4[[2, 6]].cpu(), tol)
    assert torch.equal(call([2]), eps1), "the same call twice must be bit-identical"
    eps2 = call([2, 3])
    _check(f"{tag}: one sample alone vs inside a batch of 2 (one UNet call)", eps1.cpu(), eps2[[0, 2]].cpu(), tol)
    # the other samples' VALUES do not matter: the batch of 4 with samples 0, 1, 3 replaced by noise gives sample 2 the same bits
    keep = x.clone()
    x[[0, 1, 3]] = torch.randn((3, 9, 64, 64), device=cuda)
    eps4b = call([0, 1, 2, 3])
    x.copy_(keep)
    assert torch.equal(eps4b[[2, 6]], eps4[[2, 6]]), "a sample's result changed with the VALUES of its batch mates"


def test_benchmarked_call_path_vs_oracle_bf16_and_fp8(engine, cuda):
    """EXACTLY the call the benchmark replays — sampling._Stepper.step's forward_nhwc: 64x64 latents, 8 samples (batch 4 with CFG),
    fused text cross-attention on folded tables, the unconditional half on the zero-context shortcut, no attention maps — against
    the fp32 CPU oracle on 2 + 2 of the samples: the bf16 path (stated 2e-2) and config #5's MX8 linears, without and with the e4m3 attention (stated TOL_CFG5 = 6e-2 against the
    reference arithmetic and against the bf16 path).  ``unet(x, t, ctx)`` — what the other single-call tests go through — takes the
    map-emitting xattn chain instead of this path."""
    import sgm.modules.hipnn as H
    from oracle import nets, spec
    from udifftext_amd import ops, synth
    torch.manual_seed(31)
    B = 4
    le = engine.conditioner.embedders[0]
    ctx = le(synth.synthetic_batch(B, 512, 512, 9, seed=6)["label"])
    tctx = torch.cat([torch.zeros_like(ctx), ctx])
    x = torch.randn((2 * B, 9, 64, 64), device=cuda)
    ts = torch.full((2 * B,), 441, device=cuda)
    unet = engine.model.diffusion_model
    ops.WORK_COUNTER = {}
    ref_bf16 = _sampler_call(unet, x, ts, tctx, B)
    assert ops.WORK_COUNTER.get("gemm_fp8_launches", 0) == 0
    prev, prev_a8 = H.FP8_LINEARS, H.FP8_ATTENTION
    H.FP8_LINEARS, H.FP8_ATTENTION = True, False             # the linears-only leg: attn1 stays on the bf16 flash kernel
    try:
        ops.WORK_COUNTER = {}
        eps = _sampler_call(unet, x, ts, tctx, B)
        n8 = ops.WORK_COUNTER.get("gemm_fp8_launches", 0)
        assert ops.WORK_COUNTER.get("attn_fp8_launches", 0) == 0 and ops.WORK_COUNTER.get("attn_launches", 0) == 16
    finally:
        H.FP8_LINEARS, H.FP8_ATTENTION = prev, prev_a8
        ops.WORK_COUNTER = None
    assert n8 == 11 * 5, f"{n8} MX8 GEMM launches"
    # ... and with attn1's Q K^T / P V on e4m3 operands too (hipnn.FP8_ATTENTION): all 16 blocks, the 320-channel level included
    H.FP8_LINEARS, prev_a8 = True, H.FP8_ATTENTION
    H.FP8_ATTENTION = True
    try:
        ops.WORK_COUNTER = {}
        eps_a8 = _sampler_call(unet, x, ts, tctx, B)
        na8, n8b = ops.WORK_COUNTER.get("attn_fp8_launches", 0), ops.WORK_COUNTER.get("gemm_fp8_launches", 0)
        assert ops.WORK_COUNTER.get("attn_launches", 0) == 0
    finally:
        H.FP8_LINEARS, H.FP8_ATTENTION = prev, prev_a8
        ops.WORK_COUNTER = None
    assert na8 == 16 and n8b == 11 * 5, (na8, n8b)
    pick = [0, 3, 4, 7]
    sd = {k: v.detach().float().cpu() for k, v in engine.state_dict().items() if k.startswith("model.")}
    with torch.no_grad():
        ref = nets.unet_forward(sd, x[pick].cpu(), ts[pick].cpu(), tctx[pick].float().cpu(), spec.EngineConfig().unet)
    _check("benchmarked call path (fused t_attn, zero-context rows) bf16 vs oracle", ref_bf16[pick].cpu(), ref, 2e-2, 8e-2)
    _check("benchmarked call path with MX8 linears ONLY (bf16 attention) vs oracle", eps[pick].cpu(), ref, TOL_CFG5)
    _check("benchmarked call path with MX8 linears ONLY vs the bf16 path", eps.cpu(), ref_bf16.cpu(), TOL_CFG5)
    _check("benchmarked call path with MX8 linears + e4m3 attention vs oracle", eps_a8[pick].cpu(), ref, TOL_CFG5)
    _check("benchmarked call path with MX8 linears + e4m3 attention vs the bf16 path", eps_a8.cpu(), ref_bf16.cpu(), TOL_CFG5)
    assert not torch.equal(eps_a8, eps), "the two config-#5 legs ran the same attention"


def test_fixed_scale_of_v_in_the_e4m3_attention_neither_saturates_nor_wastes_its_range(engine, cuda, monkeypatch):
    """config #5's e4m3 attention writes v with ONE data-free multiplier per layer (hipnn.v_fixed_mul: 448 at 12 ||w_j||_2 + |c_j| of the
    widest output row; the emitting epilogue saturates).  On the benchmarked call path, for every one of the 16 attn1 layers: no
    element of v sits on the saturation value, and the largest |v| of the layer uses at least the format's upper 8 binades (an e4m3
    normal has 14) — so the coarse steps of the subnormal range are >= 6 binades below the layer's maximum."""
    import sgm.modules.hipnn as H
    from udifftext_amd import ops, synth
    torch.manual_seed(33)
    B = 2
    le = engine.conditioner.embedders[0]
    ctx = le(synth.synthetic_batch(B, 512, 512, 9, seed=8)["label"])
    tctx = torch.cat([torch.zeros_like(ctx), ctx])
    x = torch.randn((2 * B, 9, 64, 64), device=cuda)
    ts = torch.full((2 * B,), 300, device=cuda)
    seen = []
    real = ops.attention_mx8

    def spy(qkv, batch, heads, scale, v_mul, **kw):
        C = heads * 64
        v = qkv.data[:, 2 * C:3 * C]
        mag = (v & 0x7F).to(torch.int32)
        seen.append((C, v_mul, float((mag >= 0x7E).float().mean()), int(mag.max())))
        return real(qkv, batch, heads, scale, v_mul, **kw)
    monkeypatch.setattr(H, "FP8_LINEARS", True)
    monkeypatch.setattr(H, "FP8_ATTENTION", True)
    monkeypatch.setattr(ops, "attention_mx8", spy)
    _sampler_call(engine.model.diffusion_model, x, ts, tctx, B)
    assert len(seen) == 16
    for C, v_mul, sat, top in seen:
        assert sat == 0.0, f"width {C}: {sat:.2e} of v saturated at multiplier {v_mul}"
        assert top >= (7 << 3), f"width {C}: largest |v| byte {top:#x} — the multiplier {v_mul} leaves the upper half of the range unused"
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write("fixed-scale v of the e4m3 attention: (width, multiplier, largest byte) " + " ".join(f"({c},{m:g},{t:#x})" for c, m, _, t in seen) + "\n")


@pytest.mark.parametrize("attn8", [False, True])
def test_config2_512_fifty_steps_with_fp8_linears_vs_reference_golden(engine, cuda, monkeypatch, attn8):
    """config #5's arithmetic over a whole trajectory: BASELINE config #2's image (512x512, 9 characters, CFG 5) through 50 Euler
    steps with the MX8 linears, against the REAL reference's latents after 10 / 25 / 50 steps (engine_golden_512.npz).  Stated
    tolerance: rel_rms <= TOL_CFG5 = 6e-2 at every horizon (bf16: 3e-2 stated / 1.2e-2 measured) — the e4m3 error of one call must not grow
    along the trajectory; decoded image <= 6e-2."""
    import sgm.modules.hipnn as H
    from udifftext_amd import config as C, pipeline, synth
    g12 = np.load(os.path.join(GOLD, "engine_golden_512.npz"))
    batch = synth.synthetic_batch(1, 512, 512, 9, seed=12)
    torch.manual_seed(1234)
    batch, buc = pipeline.prepare_batch(batch, cuda)
    c, uc = engine.conditioner.get_unconditional_conditioning(batch, batch_uc=buc, force_uc_zero_embeddings=["label"])
    monkeypatch.setattr(H, "FP8_LINEARS", True)
    monkeypatch.setattr(H, "FP8_ATTENTION", attn8)            # (True: attn1's Q K^T and P V on e4m3 operands as well)
    tag = " + e4m3 attention" if attn8 else ""
    cfgs = C.default_runtime_config(steps=50, batch_size=1, noise_iters=0)
    from sgm.modules.diffusionmodules.sampling import _Stepper
    sampler = pipeline.init_sampling(50, 5.0, cuda)
    torch.manual_seed(512)
    x0 = sampler.get_init_noise(cfgs, engine, cond=c, batch=batch, uc=uc)
    np.testing.assert_array_equal(x0.cpu().numpy(), g12["g12_x0"])
    sig = sampler._host_sigmas()
    z = x0.clone().float() * (1.0 + sig[0] ** 2.0) ** 0.5
    st = _Stepper(engine, c, uc, 1, z.shape[2:], 5.0)
    for i in range(50):
        st.step(z, sig[i], sig[i + 1])
        if i + 1 in (10, 25, 50):
            _check(f"512x512 with MX8 linears{tag}: latent after {i + 1} steps vs reference", z.cpu(), g12[f"g12_latent_{i + 1}"], TOL_CFG5)
    st.check()
    # (the sampler's own entry point, hipGraph replay included, must give the same trajectory)
    z2 = sampler(engine, x0.clone(), cond=c, batch=batch, uc=uc)
    _check(f"512x512 with MX8 linears{tag}: graph-replayed sampler vs eager steps", z2.cpu(), z.cpu(), 1e-6)
    dec = engine.decode_first_stage(z)
    _check(f"512x512 with MX8 linears{tag}: decoded image of the 50-step latent vs reference", dec[:, :, ::8, ::8].cpu(), g12["g12_decoded_sub"], TOL_CFG5)


def test_noise_search_at_benchmarked_latent_size_vs_oracle(engine, cuda):
    """the reference-default noise search (configs/test.yaml:14) at 64x64 latents with 4 candidates: every candidate's
    local attention loss against the fp32 CPU oracle's (2 Euler steps + get_min_local_loss per candidate), the same
    candidate wins, and the draws come from the CPU generator in the reference's order"""
    from oracle import sampling as osamp, spec
    from udifftext_amd import config as C, pipeline, synth
    batch = synth.synthetic_batch(1, 512, 512, 9, seed=21)
    torch.manual_seed(1234)
    batch, buc = pipeline.prepare_batch(batch, cuda)
    c, uc = engine.conditioner.get_unconditional_conditioning(batch, batch_uc=buc, force_uc_zero_embeddings=["label"])
    sampler = pipeline.init_sampling(50, 5.0, cuda)
    cfgs = C.default_runtime_config(steps=50, batch_size=1, noise_iters=4)
    import io, contextlib
    torch.manual_seed(2024)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        x0 = sampler.get_init_noise(cfgs, engine, cond=c, batch=batch, uc=uc)
    line = [l for l in buf.getvalue().splitlines() if l.startswith("Init local loss")][0]
    best, worst = float(line.split("Best")[1].split("Worst")[0]), float(line.split("Worst")[1])
    # oracle: the same conditioning tensors (so only the search is under test), the same CPU draws
    sd = {k: v.detach().float().cpu() for k, v in engine.state_dict().items()}
    cpu = lambda d: {k: (v.float().cpu() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}
    torch.manual_seed(2024)
    with torch.no_grad():
        want, scores = osamp.get_init_noise(sd, spec.EngineConfig(), (1, 4, 64, 64), cpu(c), cpu(uc), cpu(batch), 4, 5.0)
    assert len(scores) == 4
    np.testing.assert_array_equal(x0.cpu().numpy(), want.numpy())                         # same winner, bit-exact draw
    _check("noise search: best local loss vs oracle", torch.tensor([best]), torch.tensor([min(scores)]), 3e-2)
    _check("noise search: worst local loss vs oracle", torch.tensor([worst]), torch.tensor([max(scores)]), 3e-2)


def test_noise_search_batched_candidates_match_one_at_a_time(engine, cuda, monkeypatch):
    """the candidates of the noise search run as extra batch entries of shared UNet calls (up to 16 samples per call); every
    candidate's score must equal the one-candidate-at-a-time score (a sample's result does not depend on its batch mates, up to
    the launch plan's fp32 summation order), the same CPU draws are consumed, and the generator ends in the same state.
    Batch of 2 images x 5 candidates at 32x32 latents: chunks of 5 candidates = 10 samples per call"""
    from sgm.modules.diffusionmodules import sampling as S
    from udifftext_amd import config as C, pipeline, synth
    batch = synth.synthetic_batch(2, 256, 256, 6, seed=31)
    torch.manual_seed(77)
    batch, buc = pipeline.prepare_batch(batch, cuda)
    c, uc = engine.conditioner.get_unconditional_conditioning(batch, batch_uc=buc, force_uc_zero_embeddings=["label"])
    sampler = pipeline.init_sampling(50, 5.0, cuda)
    cfgs = C.default_runtime_config(steps=50, batch_size=2, noise_iters=5)
    import io, contextlib

    def run(flag):
        monkeypatch.setattr(S, "NOISE_BATCH", flag)
        torch.manual_seed(4321)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            x0 = sampler.get_init_noise(cfgs, engine, cond=c, batch=batch, uc=uc)
        line = [l for l in buf.getvalue().splitlines() if l.startswith("Init local loss")][0]
        return x0, float(line.split("Best")[1].split("Worst")[0]), float(line.split("Worst")[1]), torch.rand(1).item()

    xa, ba, wa, ra = run(True)
    xb, bb, wb, rb = run(False)
    assert ra == rb                                             # K + 1 draws either way
    assert abs(ba - bb) <= 2e-3 * abs(bb) + 1e-6 and abs(wa - wb) <= 2e-3 * abs(wb) + 1e-6, (ba, bb, wa, wb)
    assert xa.shape == xb.shape == (2, 4, 32, 32)
    # (the arg-min between candidates whose scores differ by less than the bf16 noise may legitimately differ; with these
    #  seeds the winners are separated by > 1 % and must agree)
    assert torch.equal(xa, xb)


def test_predict_many_in_flight_with_noise_search(engine, cuda):
    """three batches in flight WITH the noise search (noise_iters > 0: the init-noise stepper of every lane plans its
    stream-K launches for the lane's share of the CUs — ADVICE round 2): same frames as predict() one batch at a time.
    One candidate per image (noise_iters 1: the two scoring UNet calls with attention maps run, but there is no arg-min
    between candidates whose scores differ by 1e-4 with random weights — that choice legitimately flips with the launch plan)"""
    from udifftext_amd import config as C, pipeline, synth
    cfgs = C.default_runtime_config(steps=3, batch_size=1, noise_iters=1)
    batches = [synth.synthetic_batch(1, 256, 256, 4, seed=40 + i) for i in range(3)]
    sampler = pipeline.init_sampling(3, 5.0, cuda)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        torch.manual_seed(5)
        seq = [pipeline.predict(cfgs, engine, sampler, {k: (v.clone() if isinstance(v, torch.Tensor) else list(v)) for k, v in b.items()}, cuda)
               for b in batches]
        torch.manual_seed(5)
        many = pipeline.predict_many(cfgs, engine, sampler, batches, cuda, in_flight=3, fuse=1)
    for i, ((s1, z1), (s2, z2)) in enumerate(zip(seq, many)):
        _check(f"in flight + noise search, batch {i}: latent vs one at a time", z2.cpu(), z1.cpu(), 2e-2)
        assert torch.isfinite(s2).all()


def test_checkpoint_load_prepare_free_masters_matches_goldens(engine, cond256, eg, cuda, tmp_path):
    """SURVEY §8f-3 on the GPU: write the engine's 1330 keys as .safetensors, build a FRESH engine, init_from_ckpt,
    prepare(free_masters=True, dedup_vae=True) — packed layouts only, fp32 masters released — and re-run the reference
    golden checks (VAE G5, UNet call G7, 10-step trajectory G9).  prepare() runs with UDT_FP8 on: the e4m3 layouts of config #5
    are built and frozen next to the bf16 ones, and the packed-only engine serves BOTH arithmetics."""
    import sgm.modules.hipnn as H
    from safetensors.torch import save_file
    from udifftext_amd import config as C, ops, pipeline
    path = os.path.join(str(tmp_path), "engine.safetensors")
    save_file({k: v.detach().cpu().contiguous() for k, v in engine.state_dict().items()}, path)
    from sgm.util import instantiate_from_config, skip_param_init
    with skip_param_init():                                 # uninitialised parameters: everything must come from the file
        fresh = instantiate_from_config(C.default_model_config().model)
    fresh.to(cuda).eval()
    fresh.freeze()
    missing, unexpected = fresh.init_from_ckpt(path)
    assert not missing and not unexpected
    prev_fp8 = H.FP8_LINEARS
    H.FP8_LINEARS = True
    try:
        rep = fresh.prepare(free_masters=True, dedup_vae=True)
    finally:
        H.FP8_LINEARS = prev_fp8
    # (the name-keyed synthetic recipe gives the twin autoencoders DIFFERENT weights, so nothing is deduplicated here; the
    #  dedup itself is covered by tests/test_dropin_cpu.py)
    assert rep["vae_deduplicated"] == 0 and rep["freed_bytes"] > 4e9
    assert fresh.model.diffusion_model.input_blocks[1][0].in_layers[2].weight.numel() == 0          # masters are gone
    g = torch.Generator().manual_seed(5)
    img = torch.rand((1, 3, 64, 64), generator=g) * 2 - 1
    mom = fresh.first_stage_model.encode_moments(img.to(cuda))
    _check("ckpt -> prepare(free_masters): VAE moments vs reference", ops.nhwc_to_nchw(mom, 8).cpu(), eg["g5_moments"], 2e-2, 8e-2)
    batch, _, _ = cond256
    torch.manual_seed(1234)
    b2, buc = pipeline.prepare_batch({k: (v.clone() if isinstance(v, torch.Tensor) else list(v)) for k, v in batch.items()}, cuda)
    c, uc = fresh.conditioner.get_unconditional_conditioning(b2, batch_uc=buc, force_uc_zero_embeddings=["label"])
    _check("ckpt -> prepare(free_masters): conditioner c.concat vs reference", c["concat"].cpu(), eg["g6_c_concat"], 2e-2, 8e-2)
    x7 = torch.from_numpy(eg["g7_x"]).to(cuda)
    ucc, cc = torch.from_numpy(eg["g6_uc_concat"]).to(cuda), torch.from_numpy(eg["g6_c_concat"]).to(cuda)
    tctx = torch.cat([torch.zeros((1, 12, 2048), device=cuda), fresh.conditioner.embedders[0](batch["label"])])
    xin = torch.cat([torch.cat([x7, x7]), torch.cat([ucc, cc])], dim=1)
    eps = fresh.model.diffusion_model(xin, timesteps=torch.tensor([999, 999], device=cuda), t_context=tctx)
    _check("ckpt -> prepare(free_masters): UNet eps vs reference", eps.cpu(), eg["g7_eps"], 2e-2, 8e-2)
    H.FP8_LINEARS = True
    try:
        ops.WORK_COUNTER = {}
        eps8 = _sampler_call(fresh.model.diffusion_model, xin, torch.tensor([999, 999], device=cuda), tctx, 1)
        n8 = ops.WORK_COUNTER.get("gemm_fp8_launches", 0)
    finally:
        H.FP8_LINEARS = prev_fp8
        ops.WORK_COUNTER = None
    assert n8 == 52
    _check("ckpt -> prepare(free_masters): UNet eps with MX8 linears vs reference", eps8.cpu(), eg["g7_eps"], TOL_CFG5)
    sampler = pipeline.init_sampling(10, 5.0, cuda)
    cfgs = C.default_runtime_config(steps=10, batch_size=1, noise_iters=0)
    torch.manual_seed(99)
    x0 = sampler.get_init_noise(cfgs, fresh, cond=c, batch=b2, uc=uc)
    z = sampler(fresh, x0.clone(), cond=c, batch=b2, uc=uc)
    _check("ckpt -> prepare(free_masters): 10-step latent vs reference", z.cpu(), eg["g9_latent"], 6e-2)
    del fresh
    torch.cuda.empty_cache()


def test_unet_call_at_768_batch8_vs_oracle(engine, cuda):
    """BASELINE config #4 at its STATED batch: 768x768 -> 96x96 latents, batch 8 -> 16 samples per UNet call (uc half first),
    12-character contexts — the tile / split-K / wide-convolution plans of that call (1152 convolution tiles at the first
    level, 9216-token self-attention on 16 samples) — against the fp32 CPU oracle on one unconditional and one conditional
    sample (samples are independent: per-sample GroupNorm / LayerNorm / attention)"""
    from oracle import nets, spec
    from udifftext_amd import synth
    torch.manual_seed(41)
    B = 8
    le = engine.conditioner.embedders[0]
    ctx = le(synth.synthetic_batch(B, 768, 768, 12, seed=9)["label"])
    tctx = torch.cat([torch.zeros_like(ctx), ctx])
    x = torch.randn((2 * B, 9, 96, 96), device=cuda)
    ts = torch.full((2 * B,), 521, device=cuda)
    eps = engine.model.diffusion_model(x, timesteps=ts, t_context=tctx)
    assert eps.shape == (2 * B, 4, 96, 96)
    pick = [3, 12]
    sd = {k: v.detach().float().cpu() for k, v in engine.state_dict().items() if k.startswith("model.")}
    with torch.no_grad():
        ref = nets.unet_forward(sd, x[pick].cpu(), ts[pick].cpu(), tctx[pick].float().cpu(), spec.EngineConfig().unet)
    _check("UNet eps at 96x96 latents, 16 samples (config #4 call, batch 8) vs oracle", eps[pick].cpu(), ref, 2e-2, 8e-2)


def test_768_batch8_predict_properties(engine, cuda):
    """config #4 end to end at its stated batch (768x768, 12 characters, batch 8, 2 steps): shapes, range, determinism, and
    image independence (image 5 of the batch == the same image sampled in a batch of its own with the same noise)"""
    from udifftext_amd import config as C, parallel, pipeline, synth
    sampler = pipeline.init_sampling(2, 5.0, cuda)
    cfgs = C.default_runtime_config(steps=2, batch_size=8, noise_iters=0)
    gb = synth.synthetic_batch(8, 768, 768, 12, seed=3)
    seeds = [parallel.image_seed(99, i) for i in range(8)]
    outs = [pipeline.predict_many(cfgs, engine, sampler, [{k: (v.clone() if isinstance(v, torch.Tensor) else list(v)) for k, v in gb.items()}],
                                  cuda, in_flight=1, fuse=1, image_seeds=[seeds])[0] for _ in range(2)]
    (s1, z1), (s2, z2) = outs
    assert s1.shape == (8, 3, 768, 768) and z1.shape == (8, 4, 96, 96)
    assert torch.isfinite(s1).all() and float(s1.min()) >= 0.0 and float(s1.max()) <= 1.0
    assert torch.equal(z1, z2) and torch.equal(s1, s2)
    cfg1 = C.default_runtime_config(steps=2, batch_size=1, noise_iters=0)
    one = pipeline.predict_many(cfg1, engine, sampler, [parallel.slice_batch(gb, 5, 6)], cuda, in_flight=1, fuse=1, image_seeds=[seeds[5:6]])[0]
    # a batch of 1 takes other tile / split-K plans: agreement at bf16-rounding level, amplified over 2 steps
    _check("768x768: image 5 of a batch of 8 vs the same image alone (2 steps)", z1[5:6].cpu(), one[1].cpu(), 3e-2)


def _nccl_world1(cuda):
    import socket
    import torch.distributed as dist
    if dist.is_initialized():
        return dist, False
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=cuda)
    return dist, True


@pytest.mark.parametrize("n_images,fp8", [(8, False), (4, True)])
def test_predict_sharded_under_nccl_world1(engine, cuda, monkeypatch, n_images, fp8):
    """the multi-GPU entry on ONE GPU: parallel.predict_sharded under torch.distributed's nccl (= RCCL) backend with a world
    of one — per-image seeds, micro-batches of 4 on two lanes, and the all_gather_into_tensor of the frames through RCCL —
    for config #3's per-GPU share (8 images) and, with the fp8 linears, config #5's (4 images).  Frames must equal
    pipeline.predict_many on the same micro-batches and seeds (the collective of a world of one is a copy)."""
    import sgm.modules.hipnn as H
    from udifftext_amd import config as C, parallel, pipeline, synth
    monkeypatch.setattr(H, "FP8_LINEARS", fp8)
    dist, mine = _nccl_world1(cuda)
    try:
        sampler = pipeline.init_sampling(2, 5.0, cuda)
        cfgs = C.default_runtime_config(steps=2, batch_size=4, noise_iters=0)
        gb = synth.synthetic_batch(n_images, 512, 512, 9, seed=77)
        clone = lambda b: {k: (v.clone() if isinstance(v, torch.Tensor) else list(v)) for k, v in b.items()}
        frames = parallel.predict_sharded(cfgs, engine, sampler, [clone(gb)], [123], dist=dist, micro_batch=4, in_flight=2, fuse=1, device=cuda)
        assert len(frames) == 1 and frames[0].shape == (n_images, 3, 512, 512) and frames[0].is_cuda
        micro = [parallel.slice_batch(clone(gb), a, min(a + 4, n_images)) for a in range(0, n_images, 4)]
        seeds = [[parallel.image_seed(123, i) for i in range(a, min(a + 4, n_images))] for a in range(0, n_images, 4)]
        ref = pipeline.predict_many(cfgs, engine, sampler, micro, cuda, in_flight=2, fuse=1, image_seeds=seeds)
        ref = torch.cat([s for s, _ in ref], 0)
        assert torch.equal(frames[0], ref), "frames through the RCCL all-gather differ from predict_many"
        assert torch.isfinite(frames[0]).all() and float(frames[0].min()) >= 0.0 and float(frames[0].max()) <= 1.0
    finally:
        if mine:
            dist.destroy_process_group()


@pytest.mark.parametrize("switch", ["UDT_LEAN=0", "UDT_LEAN_CONV=0", "UDT_WIDE_CONV=0", "UDT_LEAN_SPLITK=1", "UDT_GN_EPI=0", "UDT_LN_GEMM=0",
                                    "UDT_DUAL_STREAM=1", "fused text cross-attention off", "UDT_FF_PROJ=0"])
def test_unet_call_with_each_switch_in_its_non_default_position(engine, cuda, monkeypatch, switch):
    """every surviving launch-path switch (the environment variables are read once at import / first use,
    so the test sets what they set) gives the same UNet call as the default path up to the other kernel family's rounding:
    8 samples at 32x32 latents (all four levels incl. 4x4 maps; 64x64 latents for UDT_WIDE_CONV — the wide kernel is only selected
    where its 256-pixel tiles fill the chip) through the call path of the sampling loop (fused text cross-attention, zero-context
    rows) — rel RMS <= 1.5e-2 between the two paths, each finite, and NOT bit-identical: the kernels are deterministic, so equal bits
    would mean the switch changed nothing on this call (round 4's versions of two cases were vacuous that way).
    UDT_GRAPHS, UDT_FP8, UDT_FUSE_GN, UDT_ATTN512 and UDT_NOISE_BATCH have tests of their own above."""
    import sgm.modules.attention as A
    import sgm.modules.diffusionmodules.sampling as S
    import sgm.modules.hipnn as H
    from udifftext_amd import lib as L, synth
    lib = L.load()
    torch.manual_seed(77)
    B = 4
    le = engine.conditioner.embedders[0]
    ctx = le(synth.synthetic_batch(B, 256, 256, 6, seed=8)["label"])
    tctx = torch.cat([torch.zeros_like(ctx), ctx])
    hw = 64 if switch.startswith("UDT_WIDE_CONV") else 32
    x = torch.randn((2 * B, 9, hw, hw), device=cuda)
    ts = torch.full((2 * B,), 333, device=cuda)
    unet = engine.model.diffusion_model
    ref = _sampler_call(unet, x, ts, tctx, B).float()
    keys = []
    try:
        if switch.startswith("UDT_LEAN=") or switch.startswith("UDT_LEAN_CONV=") or switch.startswith("UDT_WIDE_CONV=") or switch.startswith("UDT_LEAN_SPLITK="):
            key = {"UDT_LEAN": "lean", "UDT_LEAN_CONV": "lean_conv", "UDT_WIDE_CONV": "wide_conv", "UDT_LEAN_SPLITK": "lean_splitk"}[switch.split("=")[0]]
            L.check(lib.udt_debug_set(key.encode(), int(switch.split("=")[1])), "udt_debug_set")
            keys.append(key)
            if key == "lean":                                   # (UDT_LEAN=0 also switches these two off at import)
                monkeypatch.setattr(H, "LN_GEMM", False)
                monkeypatch.setattr(H, "GN_EPI", False)
        elif switch == "UDT_GN_EPI=0":
            monkeypatch.setattr(H, "GN_EPI", False)
            monkeypatch.setattr(H, "EMIT_STATS", H.FUSE_GN)
        elif switch == "UDT_LN_GEMM=0":
            monkeypatch.setattr(H, "LN_GEMM", False)
        elif switch == "UDT_DUAL_STREAM=1":
            monkeypatch.setattr(S, "DUAL_STREAM", True)
        elif switch == "UDT_FF_PROJ=0":
            monkeypatch.setattr(A, "FF_PROJ", False)          # (round 6: ff.net[2] and proj_out as two launches again)
        else:
            monkeypatch.setattr(A, "TATTN_FUSED", False)
        if switch == "UDT_DUAL_STREAM=1":
            # the switch lives in the sampler's step: one Euler step through _Stepper with and without the two-stream split
            c = {"t_crossattn": ctx, "concat": torch.randn((B, 5, 32, 32), device=cuda)}
            uc = {"t_crossattn": torch.zeros_like(ctx), "concat": c["concat"].clone()}
            outs = []
            for dual in (False, True):
                st = S._Stepper(engine, c, uc, B, (32, 32), 5.0, two_streams=dual)
                z = x[:B, :4].clone().float().contiguous()
                st.step(z, 5.0, 4.0)
                st.check()
                outs.append(z.clone())
            got, ref = outs[1], outs[0]
        elif switch == "fused text cross-attention off":
            # (prepare_fused_tattn returns no tables with the module switch off: the call takes the layernorm -> to_q -> xattn -> to_out chain)
            from udifftext_amd import ops
            from sgm.modules.diffusionmodules.openaimodel import CPAD
            t_kv = unet.project_context(tctx)
            assert all(tb is None for lst in unet.prepare_fused_tattn(t_kv) for tb in lst)
            eps = unet.forward_nhwc(ops.nchw_to_nhwc(x.float().contiguous(), CPAD), unet.time_embedding_rows(ts), t_kv, emit_maps=False,
                                    zero_ctx_rows=B, t_fused=None)
            got = ops.nhwc_to_nchw(eps, unet.out_channels).float()
        else:
            got = _sampler_call(unet, x, ts, tctx, B).float()
        torch.cuda.synchronize()
    finally:
        for k in keys:
            L.check(lib.udt_debug_set(k.encode(), -1), "udt_debug_set")
    assert torch.isfinite(got).all()
    if switch != "UDT_DUAL_STREAM=1":       # (a scheduling switch: the same kernels on two streams may well agree bit for bit)
        assert not torch.equal(got, ref), f"{switch}: bit-identical to the default path — the switch did not change the launch path of this call"
    _check(f"UNet call with {switch} vs the default path", got.cpu(), ref.cpu(), 1.5e-2)


def test_conditioner_notices_a_touched_unconditional_batch(engine, cuda):
    """pipeline.prepare_batch marks its unconditional batch as a clone of the conditional one (the masked-image encoder pass is then
    shared without a device-side comparison).  The marker records identity + version counter of every tensor: a caller that writes to
    the unconditional batch afterwards (a different masked image for uc) must get that image encoded, not the conditional one's."""
    from udifftext_amd import pipeline, synth
    emb = engine.conditioner
    torch.manual_seed(5)
    b, buc = pipeline.prepare_batch(synth.synthetic_batch(1, 256, 256, 4, seed=3), cuda)
    c0, uc0 = emb.get_unconditional_conditioning(b, batch_uc=buc, force_uc_zero_embeddings=["label"])
    torch.manual_seed(5)
    b, buc = pipeline.prepare_batch(synth.synthetic_batch(1, 256, 256, 4, seed=3), cuda)
    buc["masked"].mul_(-1.0)                                   # in place: same tensor object, bumped version counter
    c1, uc1 = emb.get_unconditional_conditioning(b, batch_uc=buc, force_uc_zero_embeddings=["label"])
    assert torch.equal(c1["concat"], c0["concat"])
    r, _ = _metrics(uc1["concat"][:, 1:].cpu(), uc0["concat"][:, 1:].cpu())
    assert r > 0.1, f"the unconditional masked latent did not change ({r}): the stale clone marker was trusted"


def test_detailed_dumps_attention_and_segment_maps(engine, cuda, tmp_path, monkeypatch):
    """configs/test.yaml `detailed: True` (reference sampling.py:384,344-346; openaimodel.py:559-591; sampling.py:254-262): at the
    middle step the text cross-attention probabilities of save_attn_layers (output_blocks.6.1) are averaged over heads, plotted,
    and the label's per-character maps saved as temp/seg_map/seg_<name>.npy; the latent is the plain sampler's up to the two
    text-attention forms' rounding on that one step."""
    from udifftext_amd import config as C, pipeline, synth
    monkeypatch.chdir(tmp_path)
    steps = 6
    batch = synth.synthetic_batch(1, 256, 256, 4, seed=9)
    label = batch["label"][0]
    sampler = pipeline.init_sampling(steps, 5.0, cuda)
    torch.manual_seed(11)
    s0, z0 = pipeline.predict(C.default_runtime_config(steps=steps, batch_size=1, noise_iters=0),
                              engine, sampler, {k: (v.clone() if isinstance(v, torch.Tensor) else list(v)) for k, v in batch.items()})
    torch.manual_seed(11)
    s1, z1 = pipeline.predict(C.default_runtime_config(steps=steps, batch_size=1, noise_iters=0, detailed=True),
                              engine, sampler, {k: (v.clone() if isinstance(v, torch.Tensor) else list(v)) for k, v in batch.items()})
    _check("detailed=True latent vs the plain sampler", z1.cpu(), z0.cpu(), 1.5e-2)
    name = batch["name"][0]
    seg = np.load(tmp_path / "temp" / "seg_map" / f"seg_{name}.npy")
    assert seg.shape == (len(label), 16, 16) and np.isfinite(seg).all()        # (output_blocks.6.1: the 2x-downsampled level of a 32x32 latent)
    assert (tmp_path / "temp" / "attn_map" / f"attn_map_{name}.png").stat().st_size > 1000
    # the maps are softmax probabilities over the 12 context tokens averaged over heads and layers: every pixel's 12 maps sum to 1,
    # so the label's first characters carry a positive share everywhere
    unet = engine.model.diffusion_model
    full = unet.save_attn_map(save_name="again", tokens=label, out_dir=str(tmp_path / "again"))
    assert full.shape == (12, 16, 16)
    np.testing.assert_allclose(full.sum(axis=0), 1.0, atol=2e-3)
    np.testing.assert_array_equal(full[:len(label)], seg)
    # against the cached probabilities of the configured layer themselves
    items = [it for it in unet.attn_map_cache if it["name"].startswith("output_blocks.6.1") and it["name"].endswith("t_attn")]
    assert len(items) == 1 and items[0]["size"] == 16
    m = items[0]["attn_map"].float().reshape(-1, items[0]["heads"], 256, 12).mean(dim=1)[-1].t().reshape(12, 16, 16).cpu().numpy()
    np.testing.assert_allclose(full, m, atol=1e-6)
