"""Attend-and-excite on the HIP path: the dX-only backward kernels (csrc/backward.hip) against torch.autograd on the same
operands, the backward-data of linears / convolutions through the forward kernels on re-packed weights, a ResBlock + SpatialTransformer
slice against the CPU oracle's autograd (oracle/backward.py), and the whole UNet's d local_loss / d x against the gradient
torch.autograd.grad returned inside the REAL reference's EulerEDMSampler.attend_and_excite (tests/golden/aae_golden.npz, G13).
``pytest -m gpu``; everything goes through the C ABI.

Stated tolerances (error rms / reference rms): kernels on given operands 1.5e-2 (bf16 storage of inputs and results, fp32 arithmetic;
the attention backward also rounds P and dS to bf16 for the MFMAs: 2e-2); the two-block slice 3e-2; the whole UNet's gradient 3e-2
(35 blocks forward AND backward in bf16; the forward alone is 1.4e-2).

Reference: sgm/modules/diffusionmodules/sampling.py:233-252, loss.py:192-235, attention.py:140-175,236-248,286-339,375-411,
openaimodel.py:99-101,139-146,183-250.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.txt")
TOL_OP, TOL_ATTN, TOL_SLICE, TOL_UNET = 1.5e-2, 2e-2, 3e-2, 3e-2


def _rel(got, ref):
    got, ref = torch.as_tensor(got).double().cpu(), torch.as_tensor(ref).double().cpu()
    return ((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-300)).item()


def _check(name, got, ref, tol):
    r = _rel(got, ref)
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(f"{name:55s} rel_rms {r:.3e} (tol {tol:.1e})\n")
    assert r <= tol, f"{name}: rel_rms {r:.3e} > {tol}"


@pytest.fixture(scope="module")
def env(cuda):
    import udifftext_amd  # noqa: F401
    from udifftext_amd import backward, lib, ops, packing, synth
    assert lib.load().udt_device_arch_ok() == 1
    torch.set_grad_enabled(False)

    class Env:
        pass
    Env.ops, Env.bw, Env.packing, Env.synth, Env.dev = ops, backward, packing, synth, cuda
    return Env


def _bf(t):
    return t.bfloat16().float()


# ------------------------------------------------------------------------------------------------ kernels vs autograd
@pytest.mark.parametrize("B,H,N", [(2, 5, 256), (1, 5, 200), (2, 10, 1024), (1, 20, 64), (1, 5, 40), (1, 5, 4096), (5, 5, 4096)])
def test_flash_attention_backward_vs_autograd(env, B, H, N):
    g = torch.Generator().manual_seed(N)
    C = H * 64
    qkv = _bf(torch.randn((B, N, 3 * C), generator=g)).to(env.dev)
    d_o = _bf(torch.randn((B, N, C), generator=g) * 0.1).to(env.dev)
    scale = 64 ** -0.5
    qb = qkv.bfloat16().contiguous()
    o = env.ops.attention_rowv(qb[..., :C], qb[..., C:2 * C], qb[..., 2 * C:], H, scale)
    with torch.enable_grad():
        t = qkv.clone().requires_grad_(True)
        q, k, v = (t[..., i * C:(i + 1) * C].reshape(B, N, H, 64).permute(0, 2, 1, 3) for i in range(3))
        ref_o = (torch.softmax(q @ k.transpose(-1, -2) * scale, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B, N, C)
        (ref,) = torch.autograd.grad((ref_o * d_o).sum(), [t])
    got = env.ops.attention_bwd(qb, o, d_o.bfloat16().contiguous(), H, scale)
    for i, nm in enumerate("qkv"):
        _check(f"attention backward d{nm} B{B} H{H} N{N}", got[..., i * C:(i + 1) * C], ref[..., i * C:(i + 1) * C], TOL_ATTN)
    assert torch.equal(got, env.ops.attention_bwd(qb, o, d_o.bfloat16().contiguous(), H, scale)), "not deterministic"


@pytest.mark.parametrize("B,H,N,L,use_dp,use_do", [(2, 5, 256, 12, True, True), (1, 10, 300, 12, True, False),
                                                    (2, 5, 64, 4, False, True), (1, 5, 64, 1, True, True)])
def test_text_cross_attention_backward_vs_autograd(env, B, H, N, L, use_dp, use_do):
    g = torch.Generator().manual_seed(7 * N + L)
    C = H * 64
    q = _bf(torch.randn((B, N, C), generator=g)).to(env.dev)
    kv = _bf(torch.randn((B, L, 2 * C), generator=g)).to(env.dev)
    d_o = _bf(torch.randn((B, N, C), generator=g)).to(env.dev)
    d_p = torch.randn((B * H, N, L), generator=g).to(env.dev)
    scale = 64 ** -0.5
    kvb = kv.bfloat16().contiguous()
    probs = torch.empty((B * H, N, L), dtype=torch.float32, device=env.dev)
    env.ops.xattention(q.bfloat16().contiguous(), kvb[..., :C], kvb[..., C:], H, 64, scale, probs=probs)
    with torch.enable_grad():
        t = q.clone().requires_grad_(True)
        qh = t.reshape(B, N, H, 64).permute(0, 2, 1, 3)
        kh = kv[..., :C].reshape(B, L, H, 64).permute(0, 2, 1, 3)
        vh = kv[..., C:].reshape(B, L, H, 64).permute(0, 2, 1, 3)
        sim = qh @ kh.transpose(-1, -2) * scale
        p = sim.softmax(dim=-1) if L > 1 else sim.sigmoid()
        out = (p @ vh).permute(0, 2, 1, 3).reshape(B, N, C)
        total = (out * d_o).sum() * (1.0 if use_do else 0.0) + (p.reshape(B * H, N, L) * d_p).sum() * (1.0 if use_dp else 0.0)
        (ref,) = torch.autograd.grad(total, [t])
    got = env.ops.xattention_bwd(kvb[..., :C], kvb[..., C:], probs, d_p if use_dp else None,
                                 d_o.bfloat16().contiguous() if use_do else None, H, scale)
    _check(f"text cross-attention backward B{B} H{H} N{N} L{L} dP{int(use_dp)} dO{int(use_do)}", got, ref, TOL_OP)


@pytest.mark.parametrize("B,heads,size", [(1, 5, 16), (2, 5, 32), (3, 10, 16)])
def test_local_loss_backward_vs_autograd(env, B, heads, size):
    from oracle import sampling as osamp
    g = torch.Generator().manual_seed(size + B)
    n, L, seg_l = size * size, 12, 12
    probs = torch.softmax(torch.randn((B * heads, n, L), generator=g) * 2.0, dim=-1)
    mask = (torch.rand((B, 1, 64, 64), generator=g) > 0.5).float()
    seg = torch.zeros((B, seg_l))
    seg[:, :5] = 1.0
    gk = osamp.gaussian_kernel(3, 1.0, 12)
    with torch.enable_grad():
        t = probs.clone().requires_grad_(True)
        ll = osamp.min_local_loss([{"name": "x.t_attn", "heads": heads, "size": size, "attn_map": t}], mask, seg, gk, 1)
        (ref,) = torch.autograd.grad(ll.sum(), [t])
    dp = torch.zeros_like(probs).to(env.dev)
    loss = torch.zeros((B,), device=env.dev)
    env.ops.local_loss_bwd(probs.to(env.dev), mask.to(env.dev), seg.to(env.dev), gk[0, 0].reshape(9).contiguous().to(env.dev), dp, loss,
                           heads, size, 1.0)
    assert torch.allclose(loss.cpu(), ll, rtol=1e-5, atol=1e-6)
    assert torch.allclose(dp.cpu(), ref, rtol=1e-5, atol=1e-8)
    assert int((ref != 0).sum()) > 0


@pytest.mark.parametrize("rows,C", [(512, 320), (300, 640), (64, 1280)])
def test_layernorm_backward_vs_autograd(env, rows, C):
    g = torch.Generator().manual_seed(C)
    x = _bf(torch.randn((rows, C), generator=g) * 2 + 0.5).to(env.dev)
    dy = _bf(torch.randn((rows, C), generator=g)).to(env.dev)
    add = _bf(torch.randn((rows, C), generator=g)).to(env.dev)
    gamma = (1 + 0.2 * torch.randn((C,), generator=g)).to(env.dev)
    beta = (0.1 * torch.randn((C,), generator=g)).to(env.dev)
    with torch.enable_grad():
        t = x.clone().requires_grad_(True)
        (ref,) = torch.autograd.grad((F.layer_norm(t, (C,), gamma, beta, 1e-5) * dy).sum(), [t])
    _check(f"LayerNorm backward {rows}x{C}", env.ops.layer_norm_bwd(x.bfloat16(), dy.bfloat16(), gamma, 1e-5), ref, TOL_OP)
    _check(f"LayerNorm backward {rows}x{C} + add", env.ops.layer_norm_bwd(x.bfloat16(), dy.bfloat16(), gamma, 1e-5, add=add.bfloat16()),
           ref + add, TOL_OP)


@pytest.mark.parametrize("B,hw,C,silu", [(2, 16, 320, True), (1, 8, 640, False), (2, 8, 1280, True), (1, 32, 320, False), (1, 64, 320, True),
                                         (2, 16, 2560, True), (1, 3, 960, False)])
def test_groupnorm_backward_vs_autograd(env, B, hw, C, silu):
    """both forms: three launches at the forward GroupNorm's parallelism (chunk partials; the default) and one workgroup per
    (sample, group) (UDT_GN_BWD_CHUNKED=0)"""
    g = torch.Generator().manual_seed(C + hw)
    x = _bf(torch.randn((B, hw, hw, C), generator=g) * 1.5 + 0.3).to(env.dev)
    dy = _bf(torch.randn((B, hw, hw, C), generator=g)).to(env.dev)
    add = _bf(torch.randn((B, hw, hw, C), generator=g)).to(env.dev)
    gamma = (1 + 0.2 * torch.randn((C,), generator=g)).to(env.dev)
    beta = (0.1 * torch.randn((C,), generator=g)).to(env.dev)
    with torch.enable_grad():
        t = x.clone().requires_grad_(True)
        y = F.group_norm(t.permute(0, 3, 1, 2), 32, gamma, beta, 1e-5)
        y = F.silu(y) if silu else y
        (ref,) = torch.autograd.grad((y.permute(0, 2, 3, 1) * dy).sum(), [t])
    got = env.ops.group_norm_bwd(x.bfloat16(), dy.bfloat16(), gamma, beta, 32, 1e-5, silu, add=add.bfloat16())
    _check(f"GroupNorm{'+SiLU' if silu else ''} backward B{B} {hw}x{hw}x{C} + add", got, ref + add, TOL_OP)
    _check(f"GroupNorm{'+SiLU' if silu else ''} backward B{B} {hw}x{hw}x{C}, no add",
           env.ops.group_norm_bwd(x.bfloat16(), dy.bfloat16(), gamma, beta, 32, 1e-5, silu), ref, TOL_OP)
    try:
        env.ops.GN_BWD_CHUNKED = False
        got1 = env.ops.group_norm_bwd(x.bfloat16(), dy.bfloat16(), gamma, beta, 32, 1e-5, silu, add=add.bfloat16())
    finally:
        env.ops.GN_BWD_CHUNKED = True
    _check(f"GroupNorm{'+SiLU' if silu else ''} backward B{B} {hw}x{hw}x{C} + add (one workgroup per group)", got1, ref + add, TOL_OP)


def test_geglu_forward_backward_vs_autograd(env):
    g = torch.Generator().manual_seed(3)
    rows, inner = 300, 1280
    ag = _bf(torch.randn((rows, 2 * inner), generator=g) * 1.5).to(env.dev)
    dy = _bf(torch.randn((rows, inner), generator=g)).to(env.dev)
    with torch.enable_grad():
        t = ag.clone().requires_grad_(True)
        a, gt = t.chunk(2, dim=-1)
        y = a * F.gelu(gt)
        (ref,) = torch.autograd.grad((y * dy).sum(), [t])
    _check("GEGLU forward on stored pre-activations", env.ops.geglu(ag.bfloat16()), y.detach(), TOL_OP)
    _check("GEGLU backward", env.ops.geglu_bwd(ag.bfloat16(), dy.bfloat16()), ref, TOL_OP)


def test_backward_data_of_linear_and_convolutions_through_the_forward_kernels(env):
    """dX of nn.Linear / Conv2d 3x3 / 1x1 / stride-2 3x3 / nearest-x2 + 3x3 = forward launches on re-packed weights"""
    from sgm.modules import hipnn as H
    g = torch.Generator().manual_seed(11)
    dev = env.dev
    lin = H.Linear(320, 1280).to(dev)
    dy = _bf(torch.randn((512, 1280), generator=g)).to(dev)
    _check("linear backward-data", env.bw.linear_bwd(lin, dy.bfloat16()), dy @ lin.weight.float(), TOL_OP)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().bfloat16()
    for name, conv, hw_in in (("3x3", H.Conv2d(320, 640, 3, padding=1), 16), ("1x1", H.Conv2d(640, 320, 1), 16),
                              ("3x3 stride 2", H.Conv2d(320, 320, 3, stride=2, padding=1), 32)):
        conv = conv.to(dev)
        x = _bf(torch.randn((2, conv.in_channels, hw_in, hw_in), generator=g)).to(dev)
        with torch.enable_grad():
            t = x.clone().requires_grad_(True)
            y = F.conv2d(t, _bf(conv.weight.float()), None, stride=conv.stride, padding=conv.padding)
            dyc = _bf(torch.randn(y.shape, generator=g)).to(dev)
            (ref,) = torch.autograd.grad((y * dyc).sum(), [t])
        got = env.bw.down_bwd(conv, nhwc(dyc), (hw_in, hw_in)) if conv.stride == 2 else env.bw.conv_bwd(conv, nhwc(dyc))
        _check(f"convolution backward-data {name}", got.float().permute(0, 3, 1, 2), ref, TOL_OP)
    dyu = _bf(torch.randn((2, 32, 32, 320), generator=g)).to(dev)
    ref_u = dyu.reshape(2, 16, 2, 16, 2, 320).sum(dim=(2, 4))
    _check("nearest x2 upsampling backward (2x2 sums)", env.ops.sum2x2(dyu.bfloat16()), ref_u, TOL_OP)


# ------------------------------------------------------------------------------------------------ the two-block slice vs the oracle
def test_resblock_plus_spatial_transformer_slice_vs_oracle_autograd(env):
    """dL/dh0 of ResBlock -> SpatialTransformer, L = local loss of the slice's t_attn map + <G, output>, against torch.autograd on
    the CPU oracle (oracle/backward.py slice_loss_and_grad) with the same synthetic weights"""
    from oracle import backward as ob, sampling as osamp, spec
    from sgm.modules.attention import SpatialTransformer
    from sgm.modules.diffusionmodules.openaimodel import ResBlock
    dev = env.dev
    C, heads, B, hw, ted = 320, 5, 2, 16, 1280
    rb = ResBlock(C, ted, 0.0, out_channels=C)
    st = SpatialTransformer(C, heads, 64, depth=1, t_context_dim=2048, use_linear=True)
    for name, m in (("res", rb), ("st", st)):
        env.synth.fill_module_(m.eval(), prefix=f"slice.{name}.")
    with torch.no_grad():                                   # (zero_module leaves these at zero: give the branches a signal)
        for p_, scale in ((rb.out_layers[3].weight, 0.02), (st.proj_out.weight, 0.05), (st.transformer_blocks[0].t_attn.to_out[0].weight, 0.05)):
            p_.copy_(torch.randn(p_.shape, generator=torch.Generator().manual_seed(p_.numel())) * scale)
    sd = {f"res.{k}": v.detach().float().clone() for k, v in rb.state_dict().items()}
    sd.update({f"st.{k}": v.detach().float().clone() for k, v in st.state_dict().items()})
    rb, st = rb.to(dev), st.to(dev)
    g = torch.Generator().manual_seed(5)
    h0 = _bf(torch.randn((B, C, hw, hw), generator=g))
    emb = torch.randn((B, ted), generator=g)
    ctx = _bf(torch.randn((B, 12, 2048), generator=g) * 0.5)
    cot = _bf(torch.randn((B, C, hw, hw), generator=g) * 1e-3)
    mask = (torch.rand((B, 1, 128, 128), generator=g) > 0.4).float()
    seg = torch.zeros((B, 12)); seg[:, :4] = 1.0
    gk = osamp.gaussian_kernel(3, 1.0, 12)
    ll_ref, out_ref, g_ref = ob.slice_loss_and_grad(sd, "res.", "st.", h0, emb, ctx, heads, mask, seg, cot, gk, min_attn_size=1)
    # HIP: forward tape + reverse
    rb.emb_offset = 0
    emb_rows = F.linear(F.silu(emb), sd["res.emb_layers.1.weight"], sd["res.emb_layers.1.bias"]).to(dev).contiguous()
    x = h0.to(dev).permute(0, 2, 3, 1).contiguous().bfloat16()
    kv = [blk.t_attn.project_context(ctx.to(dev).bfloat16()) for blk in st.transformer_blocks]
    rec = []
    h1, b_res = env.bw.resblock_fwd(rb, x, emb_rows)
    h2, b_st = env.bw.spatial_transformer_fwd(st, h1, kv, rec, "slice.")
    _check("slice forward (tape mode) vs oracle", h2.float().permute(0, 3, 1, 2), out_ref, 2e-2)
    it = rec[0]
    it["d_probs"] = torch.zeros_like(it["attn_map"])
    loss = torch.zeros((B,), device=dev)
    env.ops.local_loss_bwd(it["attn_map"], mask.to(dev), seg.to(dev), gk[0, 0].reshape(9).contiguous().to(dev), it["d_probs"], loss,
                           heads, hw, 1.0)
    assert torch.allclose(loss.cpu(), ll_ref, rtol=2e-2, atol=1e-4), (loss.cpu(), ll_ref)
    d_h1 = b_st(cot.to(dev).permute(0, 2, 3, 1).contiguous().bfloat16())
    d_h0, _ = b_res(d_h1)
    _check("slice dL/dh0 (local loss + output cotangent) vs oracle autograd", d_h0.float().permute(0, 3, 1, 2), g_ref, TOL_SLICE)
    # the probability gradient alone (no output cotangent): the path attend-and-excite takes through its LAST counted layer
    _, _, g_ref_p = ob.slice_loss_and_grad(sd, "res.", "st.", h0, emb, ctx, heads, mask, seg, None, gk, min_attn_size=1)
    d_h0p, _ = b_res(b_st(None))
    _check("slice dL/dh0 (local loss only) vs oracle autograd", d_h0p.float().permute(0, 3, 1, 2), g_ref_p, TOL_SLICE)


# ------------------------------------------------------------------------------------------------ the whole UNet vs the real reference
@pytest.fixture(scope="module")
def engine(cuda):
    from udifftext_amd import lib, pipeline
    assert lib.load().udt_device_arch_ok() == 1
    torch.set_grad_enabled(False)
    return pipeline.build_engine(cuda)


def test_g13_attend_and_excite_gradient_and_update_vs_reference_golden(engine, env):
    """d local_loss / d x through the whole UNet (16x16 latents, B = 1) against the gradient torch.autograd.grad returned inside the
    real reference's attend_and_excite (sampling.py:233-252), then EulerEDMSampler.attend_and_excite's update itself"""
    from aae_fixture import aae_batch, aae_functional_weights
    from udifftext_amd import pipeline
    g13 = np.load(os.path.join(GOLD, "aae_golden.npz"))
    dev = env.dev
    batch = aae_batch()                      # (masks that decide the loss's hard selections: see tests/aae_fixture.py)
    torch.manual_seed(1234)
    batch, buc = pipeline.prepare_batch(batch, dev)
    c, uc = engine.conditioner.get_unconditional_conditioning(batch, batch_uc=buc, force_uc_zero_embeddings=["label"])
    _check("G13 conditioner c.concat vs reference", c["concat"].cpu(), g13["g13_c_concat"], 2e-2)
    _check("G13 conditioner c.t_crossattn vs reference", c["t_crossattn"].cpu(), g13["g13_c_txt"], 2e-2)
    # The gradient is taken at the GOLDEN's conditioning: it depends on the DIFFERENCES between the label-embedding tokens, which are
    # nearly equal vectors — with this path's own conditioning (1.1e-2 from the reference's: the bf16 label encoder) the fp32 oracle's
    # gradient itself moves by 28 % (profiles/r06_aae_debug.txt).  The conditioner has its own parity tests; this one pins the UNet's
    # reverse pass.
    c = {"concat": torch.from_numpy(g13["g13_c_concat"]).to(dev), "t_crossattn": torch.from_numpy(g13["g13_c_txt"]).to(dev)}
    x = torch.from_numpy(g13["g13_x"]).to(dev)
    sigma = torch.from_numpy(g13["g13_sigma"]).to(dev)
    sampler = pipeline.init_sampling(10, 5.0, dev)
    c_noise = sampler.get_c_noise(x, engine, sigma)
    assert torch.equal(c_noise.cpu().long(), torch.from_numpy(g13["g13_c_noise"]).long())
    unet = engine.model.diffusion_model
    loss, grad = env.bw.unet_local_loss_grad(unet, engine.loss_fn, x, c_noise.float(), c["concat"], c["t_crossattn"], batch["mask"],
                                             batch["seg_mask"])
    assert abs(float(loss[0]) - float(g13["g13_local_loss"][0])) <= 2e-2 * abs(float(g13["g13_local_loss"][0])) + 1e-4
    _check("G13 attend-and-excite gradient (whole UNet, HIP reverse pass) vs reference", grad.cpu(), g13["g13_grad"], TOL_UNET)
    alpha = float(g13["g13_alpha"][0])
    x2 = sampler.attend_and_excite(x, engine, sigma, c, batch, alpha, False, 0.0)
    step_ref = torch.from_numpy(g13["g13_x_updated"] - g13["g13_x"])
    _check("G13 attend-and-excite update x' - x vs reference", (x2 - x).cpu(), step_ref, TOL_UNET)
    # the sampler replays the evaluation as ONE hipGraph (backward.GraphedLocalLossGrad): same launches, same results, at any input
    runner = sampler._aae_runner
    assert runner is not None and runner.graph is not None
    args = (c_noise.float(), c["concat"], c["t_crossattn"], batch["mask"], batch["seg_mask"])
    for xx, ts in ((x, args[0]), (x2, args[0]), (x2 * 0.5, args[0] - 300.0)):
        l_g, g_g = runner(xx, ts, *args[1:])
        l_e, g_e = env.bw.unet_local_loss_grad(unet, engine.loss_fn, xx, ts, *args[1:])
        assert torch.equal(g_g, g_e) and torch.allclose(l_g, l_e, rtol=1e-5, atol=1e-6)
    runner.check()
    assert sampler.attend_and_excite(x, engine, sigma, c, batch, alpha, False, 0.0).equal(x2) and sampler._aae_runner is runner
    import sgm.modules.diffusionmodules.sampling as S
    try:                                                                   # UDT_AAE_GRAPH=0: eager launches, same update
        S.AAE_GRAPH = False
        s2 = pipeline.init_sampling(10, 5.0, dev)
        x3 = s2.attend_and_excite(x, engine, sigma, c, batch, alpha, False, 0.0)
        assert getattr(s2, "_aae_runner", None) is None and torch.equal(x3, x2)
    finally:
        S.AAE_GRAPH = True
    # G13s: a DENSE cotangent on every counted map (the smooth functional sum_k <R_k, map_k> / count of the real reference's maps)
    names = [str(n) for n in g13["g13_map_names"]]

    def maps_grad(rec):
        used = [it for it in rec if it["size"] >= engine.loss_fn.min_attn_size]
        assert [it["name"] for it in used] == names
        val = 0.0
        for k, it in enumerate(used):
            r = aae_functional_weights(it["attn_map"].shape, k).to(dev)
            it["d_probs"] = (r / len(used)).contiguous()
            val += float((r * it["attn_map"]).sum()) / len(used)
        assert abs(val - float(g13["g13s_value"][0])) <= 2e-2 * abs(float(g13["g13s_value"][0]))
    gs = env.bw.unet_maps_vjp(unet, x, c_noise.float(), c["concat"], c["t_crossattn"], maps_grad)
    _check("G13s dense map cotangents through the whole UNet (HIP reverse pass) vs reference", gs.cpu(), g13["g13s_grad"], TOL_UNET)
    # deterministic
    _, grad2 = env.bw.unet_local_loss_grad(unet, engine.loss_fn, x, c_noise.float(), c["concat"], c["t_crossattn"], batch["mask"],
                                           batch["seg_mask"])
    assert torch.equal(grad, grad2)


def test_reverse_pass_at_512_vs_oracle_autograd(engine, env):
    """the benchmark's latent size (64 x 64: all four UNet levels, the 4096-token flash backward with its deep load ring, the wide
    convolution as backward-data kernel): d / d x of the smooth map functional sum_k <R_k, t_attn map_k> / count, B = 1, HIP reverse pass
    against torch.autograd through the fp32 CPU oracle with the same synthetic weights"""
    from aae_fixture import aae_functional_weights
    from oracle import backward as obw, spec
    from udifftext_amd import pipeline
    dev = env.dev
    batch = env.synth.synthetic_batch(1, 512, 512, 9, seed=6)
    torch.manual_seed(17)
    batch, buc = pipeline.prepare_batch(batch, dev)
    c, uc = engine.conditioner.get_unconditional_conditioning(batch, batch_uc=buc, force_uc_zero_embeddings=["label"])
    x = torch.randn((1, 4, 64, 64), device=dev) * 3.0
    sigma = torch.full((1,), 2.5, device=dev)
    sampler = pipeline.init_sampling(10, 5.0, dev)
    c_noise = sampler.get_c_noise(x, engine, sigma)
    unet = engine.model.diffusion_model
    min_size = engine.loss_fn.min_attn_size

    def maps_grad(rec):
        used = [it for it in rec if it["size"] >= min_size]
        for k, it in enumerate(used):
            it["d_probs"] = (aae_functional_weights(it["attn_map"].shape, k).to(dev) / len(used)).contiguous()
    got = env.bw.unet_maps_vjp(unet, x, c_noise.float(), c["concat"], c["t_crossattn"], maps_grad).cpu()
    sd = {k: v.detach().float().cpu() for k, v in engine.state_dict().items()}
    cond = {"concat": c["concat"].float().cpu(), "t_crossattn": c["t_crossattn"].float().cpu()}
    _, ref = obw.maps_functional_grad(sd, spec.EngineConfig(), x.cpu(), sigma.cpu(), cond, aae_functional_weights, min_size)
    _check("reverse pass at 64x64 latents (dense map cotangents) vs oracle autograd", got, ref, 3e-2)


def test_sampling_with_attend_and_excite_runs_and_lowers_the_local_loss(engine, env):
    """aae_enabled: True through the sampler (reference sampling.py:355-420): finite latent, one local loss and one decoded
    intermediate per step, and the iterated updates at the scheduled steps push the loss DOWN relative to the plain trajectory"""
    from udifftext_amd import config as C, pipeline
    dev = env.dev
    batch = env.synth.synthetic_batch(1, 128, 128, 4, seed=14)
    torch.manual_seed(99)
    batch, buc = pipeline.prepare_batch(batch, dev)
    c, uc = engine.conditioner.get_unconditional_conditioning(batch, batch_uc=buc, force_uc_zero_embeddings=["label"])
    sampler = pipeline.init_sampling(8, 5.0, dev)
    cfgs = C.default_runtime_config(steps=8, batch_size=1, noise_iters=0)
    torch.manual_seed(5)
    x0 = sampler.get_init_noise(cfgs, engine, cond=c, batch=batch, uc=uc)
    z = sampler(engine, x0.clone(), cond=c, batch=batch, uc=uc, aae_enabled=True)
    assert bool(torch.isfinite(z).all()) and len(sampler.last_local_losses) == 8 and len(sampler.last_inters) == 8
    assert sampler.last_inters[0].shape == (128, 128, 3)
    z_plain = sampler(engine, x0.clone(), cond=c, batch=batch, uc=uc, aae_enabled=False)
    assert not torch.equal(z, z_plain)


def test_predict_with_aae_enabled_end_to_end(engine, env):
    """configs/test.yaml ``aae_enabled: True`` through pipeline.predict (= test.py:19-40): conditioner -> noise -> the sampler's
    attend-and-excite loop -> decode; frames finite and in range, and different from the plain run's"""
    from udifftext_amd import config as C, pipeline
    dev = env.dev
    sampler = pipeline.init_sampling(6, 5.0, dev)
    outs = []
    for aae in (True, False):
        cfgs = C.default_runtime_config(steps=6, batch_size=1, noise_iters=0)
        cfgs.aae_enabled = aae
        batch = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in env.synth.synthetic_batch(1, 128, 128, 4, seed=21).items()}
        torch.manual_seed(3)
        frames, z = pipeline.predict(cfgs, engine, sampler, batch, dev)
        assert frames.shape == (1, 3, 128, 128) and bool(torch.isfinite(frames).all())
        assert float(frames.min()) >= 0.0 and float(frames.max()) <= 1.0
        outs.append(z)
    assert not torch.equal(outs[0], outs[1])
    assert len(sampler.last_local_losses) == 6
