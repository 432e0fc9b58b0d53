"""The lean co-resident kernel family of csrc/lean.h (every plain / GEGLU / LayerNorm-fed linear, the 1x1 and the 3x3
stride-1 convolutions of the UNet run on it by default) against a plain PyTorch fp32 reference of the same op on the same
bf16-rounded inputs, per tile configuration (forced through ``udt_debug_set``), with and without the ticket split-K, plus
run-to-run bit-reproducibility (the split-K sums its slabs in slice order, not in arrival order).  ``pytest -m gpu``.

Reference ops: nn.Linear / GEGLU / LayerNorm→Linear of sgm/modules/attention.py:83-99,310-339, nn.Conv2d 1x1 / 3x3 and
Upsample→Conv2d of sgm/modules/diffusionmodules/openaimodel.py:96-133,262-282.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# stated tolerance: fp32-accumulated bf16 products + one bf16 rounding of the output (rel 2^-8); the LayerNorm-folded
# form additionally rounds gamma*W to bf16 and subtracts mean*colsum in fp32 (cancellation), hence the wider bound
REL_RMS = 6e-3
REL_RMS_LN = 1e-2


def _rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30)).item()


@pytest.fixture(scope="module")
def env(cuda):
    import udifftext_amd  # noqa: F401
    from udifftext_amd import lib as L, ops as O, packing as P
    lib = L.load()
    assert lib.udt_device_arch_ok() == 1, "tests expect a gfx950 device"

    class Env:
        ops, packing, GEGLU = O, P, L.GEMM_GEGLU

        @staticmethod
        def dbg(key, val):
            L.check(lib.udt_debug_set(key.encode(), int(val)), "udt_debug_set " + key)

        @staticmethod
        def reset():
            for k in ("lean", "lean_splitk", "lean_conv", "wide_conv", "rowres", "share_splitk", "lean256_lanes", "conv_n4", "wide_lanes_eff"):
                L.check(lib.udt_debug_set(k.encode(), -1), "udt_debug_set " + k)
    yield Env
    Env.reset()


def _linear_case(env, dev, M, N, K, flags, res, rpb, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn((M, K), generator=g).to(dev).bfloat16()
    w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(dev) * (1.0 + torch.arange(N, device=dev)[:, None] / N)
    b = torch.randn((N,), generator=g).to(dev)
    wp, bp = env.packing.pack_geglu(w, b) if flags & env.GEGLU else (env.packing.pack_linear(w), b)
    r = torch.randn((M, N // 2 if flags & env.GEGLU else N), generator=g).to(dev).bfloat16() if res else None
    rv = torch.randn((M // rpb, N), generator=g).to(dev) if rpb else None
    y = x.float() @ w.bfloat16().float().t() + b
    if flags & env.GEGLU:
        y = y[:, :N // 2] * F.gelu(y[:, N // 2:])
    if rv is not None:
        y = y + rv.repeat_interleave(rpb, 0)
    if r is not None:
        y = y + r.float()
    return x, wp, bp, dict(residual=r, rowvec=rv, rows_per_batch=rpb, flags=flags), y


LINEAR_CASES = [  # M, N, K, geglu, residual, rows_per_batch (time-embedding row vector)
    (4096, 320, 320, False, True, 0),        # L0 proj_out + residual (128x160 tiles when automatic)
    (4096, 960, 320, False, False, 0),       # fused q|k|v
    (2048, 2560, 320, True, False, 0),       # GEGLU, N=2*1280
    (2048, 640, 2560, False, True, 0),       # ff.net[2] + residual
    (512, 1280, 5120, False, True, 0),       # deep K: ticket split-K candidates
    (512, 10240, 1280, True, False, 0),      # L2/L3 GEGLU (256x256 tiles when automatic)
    (1000, 328, 192, False, True, 0),        # ragged M, N not a tile multiple
    (130, 136, 64, False, False, 0),         # one K tile, partial tiles both ways
    (4096, 640, 640, False, True, 1024),     # rowvec per batch (emb_layers output added per sample)
    (300, 1280, 1280, False, False, 100),
]


@pytest.mark.parametrize("cfg", [-1, 1, 6])
@pytest.mark.parametrize("case", LINEAR_CASES, ids=lambda c: "x".join(str(int(v)) for v in c))
def test_lean_linear_vs_torch(env, cuda, cfg, case):
    """cfg -1 = the library's own choice (incl. the 128x160 and 256x256 tiles), 1 / 6 = forced 4-wave 128x128 (two workgroups
    per CU) / 8-wave 256x256 (one per CU); each without and with forced split-K"""
    M, N, K, geglu, res, rpb = case
    x, wp, bp, kw, y = _linear_case(env, cuda, M, N, K, env.GEGLU if geglu else 0, res, rpb)
    try:
        env.dbg("lean", cfg)
        for sk in (-1, 3):
            env.dbg("lean_splitk", sk)
            out = env.ops.linear(x, wp, bp, **kw)
            torch.cuda.synchronize()
            e = _rel(out, y)
            assert math.isfinite(e) and e < REL_RMS, f"lean cfg {cfg} splitk {sk} {case}: rel rms {e:.3e}"
    finally:
        env.reset()


def test_lean_matches_stream_k_kernels(env, cuda):
    """same products through the 8-wave stream-K kernels (UDT_LEAN=0 path) and the lean family: both round an fp32
    accumulator once, so they agree to one bf16 ulp of the output"""
    for case in [(4096, 320, 320, False, True, 0), (512, 1280, 5120, False, True, 0), (2048, 2560, 320, True, False, 0)]:
        M, N, K, geglu, res, rpb = case
        x, wp, bp, kw, y = _linear_case(env, cuda, M, N, K, env.GEGLU if geglu else 0, res, rpb, seed=3)
        try:
            env.dbg("lean", 0)
            old = env.ops.linear(x, wp, bp, **kw)
            env.dbg("lean", -1)
            new = env.ops.linear(x, wp, bp, **kw)
        finally:
            env.reset()
        torch.cuda.synchronize()
        d = (old.float() - new.float()).abs()
        ulp = y.abs().clamp_min(1e-3) * 2.0 ** -7
        assert (d <= ulp).all(), f"{case}: {(d > ulp).sum().item()} outputs differ by more than one bf16 ulp, max {d.max().item():.3e}"


@pytest.mark.parametrize("M,N,K,geglu", [(4096, 960, 320, False), (2048, 5120, 640, True), (1024, 3840, 1280, False),
                                         (777, 2560, 320, True), (512, 1280, 1280, False), (512, 10240, 1280, True)])
def test_ln_gemm_fwd_vs_torch(env, cuda, M, N, K, geglu):
    """udt_ln_gemm_fwd: LayerNorm folded into the GEMM — LN(x) W^T = rstd (x W'^T - mean s) + c on W' = gamma∘W — against
    torch layer_norm → linear (→ GEGLU) in fp32, and against the unfused two-launch form"""
    g = torch.Generator(device="cpu").manual_seed(5)
    x = (torch.randn((M, K), generator=g) * 1.5 + 0.3).to(cuda).bfloat16()
    w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(cuda)
    b = torch.randn((N,), generator=g).to(cuda)
    gamma = (1 + 0.1 * torch.randn((K,), generator=g)).to(cuda)
    beta = (0.05 * torch.randn((K,), generator=g)).to(cuda)
    y = F.layer_norm(x.float(), (K,), gamma, beta, 1e-5) @ w.t() + b
    if geglu:
        y = y[:, :N // 2] * F.gelu(y[:, N // 2:])
    fl = env.GEGLU if geglu else 0
    wf, cf, sf = env.packing.pack_ln_linear(w, b, gamma, beta, geglu=geglu)
    try:
        for cfg in (-1, 1, 6):
            env.dbg("lean", cfg)
            out = env.ops.ln_linear(x, wf, cf, sf, flags=fl)
            torch.cuda.synchronize()
            e = _rel(out, y)
            assert math.isfinite(e) and e < REL_RMS_LN, f"ln_gemm cfg {cfg}: rel rms {e:.3e}"
    finally:
        env.reset()
    wp, bp = env.packing.pack_geglu(w, b) if geglu else (env.packing.pack_linear(w), b)
    two = env.ops.linear(env.ops.layer_norm(x, gamma, beta), wp, bp, flags=fl)
    assert _rel(out, two) < REL_RMS_LN


@pytest.mark.parametrize("M,N,K,geglu", [(32768, 2560, 320, True), (32768, 960, 320, False), (777, 2560, 320, True),
                                         (1000, 1024, 320, False), (16384, 2560, 320, True), (300, 128, 320, False),
                                         (65536, 2560, 320, True)])
def test_row_resident_ln_gemm_vs_torch_and_lean(env, cuda, M, N, K, geglu):
    """rowres.h (rgemm_kernel: the rows' A fragments resident in registers, weights streamed in 64-row chunks; forced with
    udt_debug_set("lean", 7), automatic for the 64 x 64 level's GEGLU / q|k|v projections) against torch layer_norm -> linear
    (-> GEGLU) in fp32 and against the tiled lean kernel on the same packed weights; ragged M (a partial 256-row block, a partial
    wave), one chunk, column splits (16384 rows: 4 splits of 10 chunks; 65536: 2 splits), and the automatic plan's choice"""
    g = torch.Generator(device="cpu").manual_seed(15)
    x = (torch.randn((M, K), generator=g) * 1.5 + 0.3).to(cuda).bfloat16()
    w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(cuda)
    b = torch.randn((N,), generator=g).to(cuda)
    gamma = (1 + 0.1 * torch.randn((K,), generator=g)).to(cuda)
    beta = (0.05 * torch.randn((K,), generator=g)).to(cuda)
    y = F.layer_norm(x.float(), (K,), gamma, beta, 1e-5) @ w.t() + b
    if geglu:
        y = y[:, :N // 2] * F.gelu(y[:, N // 2:])
    fl = env.GEGLU if geglu else 0
    wf, cf, sf = env.packing.pack_ln_linear(w, b, gamma, beta, geglu=geglu)
    try:
        env.dbg("lean", 7)
        out = env.ops.ln_linear(x, wf, cf, sf, flags=fl)
        torch.cuda.synchronize()
        e = _rel(out, y)
        assert math.isfinite(e) and e < REL_RMS_LN, f"row-resident ln_gemm: rel rms {e:.3e}"
        assert torch.equal(out, env.ops.ln_linear(x, wf, cf, sf, flags=fl)), "not reproducible"
        env.dbg("lean", 1)
        tiled = env.ops.ln_linear(x, wf, cf, sf, flags=fl)
        torch.cuda.synchronize()
        # the two kernels differ by fp32 rounding in front of the bf16 rounding: a few outputs land on the neighbouring bf16
        d = (out.float() - tiled.float()).abs()
        assert (d <= 2.0 ** -7 * tiled.float().abs() + 1e-6).all(), "more than one bf16 step from the tiled kernel"
        assert (d > 0).float().mean().item() < 0.02
        env.dbg("lean", -1)
        env.dbg("rowres", 0)
        off = env.ops.ln_linear(x, wf, cf, sf, flags=fl)
        env.dbg("rowres", 1)
        auto = env.ops.ln_linear(x, wf, cf, sf, flags=fl)
        torch.cuda.synchronize()
        assert _rel(off, y) < REL_RMS_LN and _rel(auto, y) < REL_RMS_LN
        if M >= 16384:
            assert torch.equal(auto, out), "the automatic plan should take the row-resident kernel here"
    finally:
        env.reset()
        env.dbg("rowres", 1)


def test_ln_gemm_constant_rows_and_large_mean(env, cuda):
    """edge cases of the folded form: a constant row (variance 0 -> rstd = 1/sqrt(eps), output = c exactly as torch gives
    beta W^T + b) and rows with a mean far from 0 (the x W'^T - mean s cancellation)"""
    M, N, K = 256, 640, 640
    g = torch.Generator(device="cpu").manual_seed(6)
    x = torch.randn((M, K), generator=g)
    x[0] = 2.0
    x[1] = 0.0
    x[2:66] += 12.0
    x = x.to(cuda).bfloat16()
    w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(cuda)
    b = torch.randn((N,), generator=g).to(cuda)
    gamma = (1 + 0.1 * torch.randn((K,), generator=g)).to(cuda)
    beta = (0.05 * torch.randn((K,), generator=g)).to(cuda)
    y = F.layer_norm(x.float(), (K,), gamma, beta, 1e-5) @ w.t() + b
    wf, cf, sf = env.packing.pack_ln_linear(w, b, gamma, beta)
    out = env.ops.ln_linear(x, wf, cf, sf)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    # constant rows: |out - c| only carries the bf16 rounding of the output
    assert (out[:2].float() - y[:2]).abs().max().item() < 2e-2
    # shifted rows: tolerance scaled by the cancellation (|mean| / std = 12)
    assert _rel(out[2:66], y[2:66]) < 3e-2
    assert _rel(out[66:], y[66:]) < REL_RMS_LN


@pytest.mark.parametrize("B,H,C1,C2,N", [(2, 64, 320, 320, 320), (2, 32, 640, 320, 640), (3, 16, 1280, 1280, 1280),
                                          (4, 8, 1280, 1280, 1280), (2, 16, 128, 64, 192), (2, 32, 320, 0, 640)])
def test_lean_conv1x1_two_sources(env, cuda, B, H, C1, C2, N):
    """skip_connection of a decoder ResBlock: Conv2d 1x1 over cat([h, skip], C) without materialising the concat
    (openaimodel.py:262-282 / th.cat at :1060)"""
    g = torch.Generator(device="cpu").manual_seed(7)
    x1 = torch.randn((B, H, H, C1), generator=g).to(cuda).bfloat16()
    x2 = torch.randn((B, H, H, C2), generator=g).to(cuda).bfloat16() if C2 else None
    w4 = (torch.randn((N, C1 + C2, 1, 1), generator=g) / math.sqrt(C1 + C2)).to(cuda)
    b = torch.randn((N,), generator=g).to(cuda)
    w = env.packing.pack_conv(w4, [C1, C2] if C2 else None)
    xin = torch.cat([x1, x2], -1) if C2 else x1
    y = F.conv2d(xin.float().permute(0, 3, 1, 2), w4.bfloat16().float(), b).permute(0, 2, 3, 1)
    try:
        for cfg in (-1, 1):
            env.dbg("lean", cfg)
            out = env.ops.conv2d(x1, w, b, ksize=1, x2=x2)
            torch.cuda.synchronize()
            e = _rel(out, y)
            assert math.isfinite(e) and e < REL_RMS, f"conv1x1 cfg {cfg}: rel rms {e:.3e}"
    finally:
        env.reset()


CONV3_CASES = [  # B, H, W, C, N, upsample, residual, rowvec
    (2, 64, 64, 320, 320, False, True, True),      # L0 ResBlock convolutions (16x8 patches, N=320: ragged N tile)
    (2, 32, 32, 640, 640, False, False, True),
    (3, 16, 16, 1280, 1280, False, True, False),
    (4, 8, 8, 1280, 1280, False, False, True),     # 8x8 maps: the 8x8-patch geometry
    (2, 24, 40, 128, 256, False, True, False),     # non-square map, partial patches on both axes
    (1, 20, 12, 64, 136, False, False, False),     # single input chunk, N not a multiple of the tile
    (2, 16, 16, 1280, 1280, True, False, False),   # Upsample (nearest x2) folded into the patch staging
    (2, 32, 32, 640, 640, True, False, False),
    (1, 12, 20, 192, 128, True, False, False),
    (2, 96, 96, 320, 320, False, True, True),      # config #4 (768 px) map size
]


@pytest.mark.parametrize("case", CONV3_CASES, ids=lambda c: "x".join(str(int(v)) for v in c))
def test_lean_conv3x3_vs_torch(env, cuda, case):
    B, H, W, C, N, ups, res, rowvec = case
    g = torch.Generator(device="cpu").manual_seed(8)
    x = torch.randn((B, H, W, C), generator=g).to(cuda).bfloat16()
    w4 = (torch.randn((N, C, 3, 3), generator=g) / math.sqrt(9 * C)).to(cuda)
    b = torch.randn((N,), generator=g).to(cuda)
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    r = torch.randn((B, Ho, Wo, N), generator=g).to(cuda).bfloat16() if res else None
    rv = torch.randn((B, N), generator=g).to(cuda) if rowvec else None
    xin = x.float().permute(0, 3, 1, 2)
    if ups:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    y = F.conv2d(xin, w4.bfloat16().float(), b, padding=1).permute(0, 2, 3, 1)
    if rv is not None:
        y = y + rv[:, None, None, :]
    if r is not None:
        y = y + r.float()
    w = env.packing.pack_conv(w4)
    kw = dict(ksize=3, upsample=ups, residual=r, rowvec=rv)
    try:
        outs = {}
        for lean_conv in (1, 0):                       # the lean patch kernel, then the 8-wave stream-K one
            env.dbg("lean_conv", lean_conv)
            out = env.ops.conv2d(x, w, b, **kw)
            torch.cuda.synchronize()
            e = _rel(out, y)
            assert math.isfinite(e) and e < REL_RMS, f"conv3x3 lean_conv={lean_conv} {case}: rel rms {e:.3e}"
            outs[lean_conv] = out
        assert _rel(outs[1], outs[0]) < REL_RMS
    finally:
        env.reset()


@pytest.mark.parametrize("case", [(2, 64, 64, 320, 4), (1, 96, 96, 320, 4), (2, 40, 24, 128, 3), (1, 13, 9, 64, 4), (3, 8, 8, 640, 4)],
                         ids=lambda c: "x".join(str(int(v)) for v in c))
def test_conv3x3_with_four_output_channels_fp32(env, cuda, case):
    """round 6, conv_n4.h: the UNet's `out` convolution (320 -> 4, reference openaimodel.py:486-490) and the VAE decoder's conv_out
    (128 -> 3, model.py:586-588) as a dot-product kernel (v_dot2_f32_bf16, halo tile in LDS, weights by scalar loads) instead of an
    MFMA tile padded to 64 columns; fp32 output; against torch and against the MFMA kernel (UDT_CONV_N4=0)"""
    B, H, W, C, N = case
    g = torch.Generator(device="cpu").manual_seed(H + C)
    x = torch.randn((B, H, W, C), generator=g).to(cuda).bfloat16()
    w4 = (torch.randn((N, C, 3, 3), generator=g) / math.sqrt(9 * C)).to(cuda)
    b = torch.randn((N,), generator=g).to(cuda)
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w4.bfloat16().float(), b, padding=1).permute(0, 2, 3, 1)
    w, bp = env.packing.pack_conv(w4), env.packing.pad_bias(b)
    assert w.shape[0] == 4
    from udifftext_amd import lib as L
    try:
        outs = {}
        for mode in (1, 0):
            env.dbg("conv_n4", mode)
            out = env.ops.conv2d(x, w, bp, ksize=3, flags=L.GEMM_OUT_F32, n_out=4)
            torch.cuda.synchronize()
            assert out.dtype == torch.float32 and out.shape == (B, H, W, 4)
            e = _rel(out[..., :N], y)
            assert math.isfinite(e) and e < REL_RMS, f"conv3x3 N=4 conv_n4={mode} {case}: rel rms {e:.3e}"
            if N < 4:
                assert not bool(out[..., N:].any())
            outs[mode] = out
        assert _rel(outs[1], outs[0]) < 1e-3          # (two fp32 summation orders of the same bf16 products)
        env.dbg("conv_n4", 1)
        assert torch.equal(env.ops.conv2d(x, w, bp, ksize=3, flags=L.GEMM_OUT_F32, n_out=4), outs[1])
    finally:
        env.reset()


def test_wide_conv_at_the_16x16_level_with_batches_in_flight(env, cuda):
    """round 6 (UDT_WIDE_LANES_EFF): a launch that shares the device with two other streams (cu_share 3) takes the wide convolution
    already when its tiles fill 3/4 of its share of the CUs — the 16 x 16 level of a UNet call (64 tiles of 256 pixels x 160 channels,
    whole tiles, no channel slices) — where the round-5 rule (85 %) took the lean kernel; a lone launch is planned as before.  Both
    plans agree with fp32; the results differ in the last bits (another tile shape), which is how the test knows the plan changed."""
    B, H, W, C, N = 8, 16, 16, 1280, 1280
    g = torch.Generator(device="cpu").manual_seed(31)
    x = torch.randn((B, H, W, C), generator=g).to(cuda).bfloat16()
    w4 = (torch.randn((N, C, 3, 3), generator=g) / math.sqrt(9 * C)).to(cuda)
    b = torch.randn((N,), generator=g).to(cuda)
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w4.bfloat16().float(), b, padding=1).permute(0, 2, 3, 1)
    w = env.packing.pack_conv(w4)
    import os, tempfile
    from udifftext_amd import lib as L
    lib = L.load()

    def run(share, eff):
        """(result, the launch's tag in the library's profiler: 'wconv3 ...' / 'lconv3 ...')"""
        env.dbg("wide_lanes_eff", eff)
        env.ops.prof_reset(); lib.udt_prof_trace(1); env.ops.prof_enable(0x3f)
        with env.ops.launch_context(cu_share=share):
            out = env.ops.conv2d(x, w, b, ksize=3)
        torch.cuda.synchronize()
        env.ops.prof_enable(0)
        path = os.path.join(tempfile.gettempdir(), "udt_wide_lanes_trace.csv")
        lib.udt_prof_dump(path.encode())
        lib.udt_prof_trace(0)
        tags = [ln.split(",", 2)[2] for ln in open(path).read().splitlines()[1:]]
        assert len(tags) == 1, tags
        return out, tags[0]
    try:
        plans = {}
        for share in (1, 3):
            for eff in (85, 70):
                out, tag = run(share, eff)
                assert _rel(out, y) < REL_RMS, (share, eff, _rel(out, y))
                plans[(share, eff)] = tag
        assert plans[(1, 85)] == plans[(1, 70)] and plans[(1, 70)].startswith("lconv3"), plans      # a lone launch: four slices -> lean
        assert plans[(3, 85)].startswith("lconv3") and plans[(3, 70)].startswith("wconv3") and "splitk=1" in plans[(3, 70)], plans
    finally:
        env.ops.prof_enable(0)
        env.reset()


def test_share_aware_split_k_plans_and_results(env, cuda):
    """round 6 (UDT_SHARE_SPLITK): a launch that shares the device with two other streams (cu_share 3) cuts K to fill ITS share of
    the workgroup slots — the deep-K few-tile GEMM and the 16x16-level convolution of the benchmarked call run whole tiles where the
    round-5 rule cut 3 slices.  Both rules agree with the fp32 reference; the results differ in the last bits (another summation
    order), which is how the test knows the plan changed; a lone launch (cu_share 1) is bit-identical under both."""
    x, wp, bp, kw, y = _linear_case(env, cuda, 2048, 1280, 5120, 0, True, 0, seed=12)
    g = torch.Generator(device="cpu").manual_seed(13)
    xc = torch.randn((8, 16, 16, 1280), generator=g).to(cuda).bfloat16()
    w4 = (torch.randn((1280, 1280, 3, 3), generator=g) / math.sqrt(9 * 1280)).to(cuda)
    bc = torch.randn((1280,), generator=g).to(cuda)
    yc = F.conv2d(xc.float().permute(0, 3, 1, 2), w4.bfloat16().float(), bc, padding=1).permute(0, 2, 3, 1)
    wc = env.packing.pack_conv(w4)
    try:
        res = {}
        env.dbg("wide_lanes_eff", 85)                 # (this test is about the lean kernels' slices: keep the convolution on them)
        for share in (3, 1):
            for rule in (1, 0):
                env.dbg("share_splitk", rule)
                with env.ops.launch_context(cu_share=share):
                    a = env.ops.linear(x, wp, bp, **kw)
                    b = env.ops.conv2d(xc, wc, bc, ksize=3)
                torch.cuda.synchronize()
                assert _rel(a, y) < REL_RMS and _rel(b, yc) < REL_RMS
                res[(share, rule)] = (a, b)
        assert not torch.equal(res[(3, 1)][0], res[(3, 0)][0]) and not torch.equal(res[(3, 1)][1], res[(3, 0)][1])
        assert torch.equal(res[(1, 1)][0], res[(1, 0)][0]) and torch.equal(res[(1, 1)][1], res[(1, 0)][1])
        assert torch.equal(res[(3, 0)][0], res[(1, 0)][0])            # (the round-5 rule ignored cu_share for these kernels)
    finally:
        env.reset()


WIDE_CASES = [  # B, H, W, C, N, residual, rowvec, forced split-K (-1 = the plan's own choice)
    (8, 64, 64, 320, 320, True, True, -1),         # L0 ResBlock convolution of the benchmarked call: 256 tiles, one per CU
    (2, 64, 64, 320, 320, False, False, -1),       # few tiles (forced onto the wide kernel)
    (8, 32, 32, 640, 640, True, True, -1),         # L1: 128 tiles -> two channel-chunk slices each
    (8, 16, 16, 1280, 1280, False, True, -1),      # L2: 64 tiles -> four slices
    (1, 16, 16, 64, 160, True, False, -1),         # one chunk, one tile
    (3, 32, 48, 192, 480, True, True, 3),          # non-square map, three chunks cut three ways
    (2, 96, 96, 320, 320, True, True, -1),         # config #4 (768 px) map
    (2, 32, 32, 960, 640, False, False, 2),        # decoder input width (concatenated skip), odd chunk count per slice
]


@pytest.mark.parametrize("case", WIDE_CASES, ids=lambda c: "x".join(str(int(v)) for v in c))
def test_wide_conv3x3_vs_torch_and_lean(env, cuda, case):
    """wide.h wconv3_kernel (256 pixels x 160 channels per workgroup, one workgroup per CU) against torch conv2d in fp32 on the
    same bf16-rounded inputs and against the lean 128-pixel kernel, incl. the ticket split-K over channel chunks, the epilogue
    statistics (GroupNorm input) and run-to-run bit-reproducibility"""
    B, H, W, C, N, res, rowvec, sk = case
    g = torch.Generator(device="cpu").manual_seed(18)
    x = torch.randn((B, H, W, C), generator=g).to(cuda).bfloat16()
    w4 = (torch.randn((N, C, 3, 3), generator=g) / math.sqrt(9 * C)).to(cuda)
    w4 = w4 * (1.0 + torch.arange(N, device=cuda)[:, None, None, None] / N)            # asymmetric in the output channel
    b = torch.randn((N,), generator=g).to(cuda)
    r = torch.randn((B, H, W, N), generator=g).to(cuda).bfloat16() if res else None
    rv = torch.randn((B, N), generator=g).to(cuda) if rowvec else None
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w4.bfloat16().float(), b, padding=1).permute(0, 2, 3, 1)
    if rv is not None:
        y = y + rv[:, None, None, :]
    if r is not None:
        y = y + r.float()
    w = env.packing.pack_conv(w4)
    kw = dict(ksize=3, residual=r, rowvec=rv)
    try:
        env.dbg("wide_conv", 0)
        lean = env.ops.conv2d(x, w, b, **kw)
        env.dbg("wide_conv", 1)
        env.dbg("lean_splitk", sk)
        out = env.ops.conv2d(x, w, b, **kw)
        again = env.ops.conv2d(x, w, b, **kw)
        st_out = env.ops.conv2d(x, w, b, colstats=True, **kw)
        torch.cuda.synchronize()
        e = _rel(out, y)
        assert math.isfinite(e) and e < REL_RMS, f"wide conv3x3 {case}: rel rms {e:.3e}"
        assert (out.float() - y).abs().max().item() <= 3e-2 * y.abs().max().item()
        assert _rel(out, lean) < REL_RMS
        assert torch.equal(out, again), "wide convolution changed bits between two launches"
        st = env.ops.gn_stats_of(st_out)
        assert st is not None and torch.equal(st_out, out)
        tot = st.data.reshape(B, st.slots_per_sample, N, 2).sum(1)
        o = out.float().reshape(B, H * W, N)
        ref = torch.stack([o.sum(1), (o * o).sum(1)], dim=-1)
        assert ((tot - ref).abs() <= 2e-4 * ref.abs() + 2e-2 * math.sqrt(H * W)).all()
        if N % 64 == 0:
            gamma = (1 + 0.1 * torch.randn((N,), generator=g)).to(cuda)
            beta = (0.1 * torch.randn((N,), generator=g)).to(cuda)
            got = env.ops.group_norm_from_stats(st_out, st, gamma, beta, 32, 1e-5, True)
            want = F.silu(F.group_norm(out.float().permute(0, 3, 1, 2), 32, gamma, beta, 1e-5)).permute(0, 2, 3, 1)
            torch.cuda.synchronize()
            assert _rel(got, want) < 6e-3
    finally:
        env.reset()


def test_lean_bit_reproducible_across_launches_and_streams(env, cuda):
    """the ticket split-K adds its slabs in slice order whatever the arrival order, and the raw s_barrier is fenced by
    s_waitcnt lgkmcnt(0): the same launch gives the same bits alone, repeated, and while a second stream keeps the device
    busy with other lean launches (the race this guards against showed up only under dual-stream load)"""
    M, N, K = 512, 1280, 5120
    x, wp, bp, kw, _ = _linear_case(env, cuda, M, N, K, 0, True, 0, seed=11)
    g = torch.Generator(device="cpu").manual_seed(12)
    cx = torch.randn((2, 32, 32, 640), generator=g).to(cuda).bfloat16()
    cw = env.packing.pack_conv((torch.randn((640, 640, 3, 3), generator=g) / math.sqrt(9 * 640)).to(cuda))
    side = torch.cuda.Stream()
    side_ws = env.ops.Workspace(cuda)
    try:
        env.dbg("lean_splitk", 4)
        first = env.ops.linear(x, wp, bp, **kw).clone()
        cfirst = env.ops.conv2d(cx, cw, None, ksize=3).clone()
        torch.cuda.synchronize()
        for it in range(6):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), env.ops.launch_context(cu_share=2, workspace=side_ws):
                for _ in range(4):
                    env.ops.conv2d(cx, cw, None, ksize=3)
            out = env.ops.linear(x, wp, bp, **kw)
            cout = env.ops.conv2d(cx, cw, None, ksize=3)
            torch.cuda.synchronize()
            assert torch.equal(out, first), f"split-K linear changed bits on repeat {it}"
            assert torch.equal(cout, cfirst), f"3x3 convolution changed bits on repeat {it}"
    finally:
        env.reset()


def test_lean_conv_plan_declines_what_it_does_not_cover(env, cuda):
    """with the lean convolution forced on, problems outside its plan (fewer than 128 output channels; the stride-2
    Downsample) still give the right product: the plan, not the caller, decides which kernel runs"""
    g = torch.Generator(device="cpu").manual_seed(13)
    for C, N, stride in [(128, 64, 1), (320, 320, 2)]:
        x = torch.randn((2, 16, 16, C), generator=g).to(cuda).bfloat16()
        w4 = (torch.randn((N, C, 3, 3), generator=g) / math.sqrt(9 * C)).to(cuda)
        y = F.conv2d(x.float().permute(0, 3, 1, 2), w4.bfloat16().float(), None, stride=stride, padding=1).permute(0, 2, 3, 1)
        try:
            env.dbg("lean_conv", 1)
            out = env.ops.conv2d(x, env.packing.pack_conv(w4), None, ksize=3, stride=stride)
            torch.cuda.synchronize()
        finally:
            env.reset()
        assert _rel(out, y) < REL_RMS


def _stats_ref(out, rows):
    """[slots, C, 2] (sum, sum of squares) over consecutive blocks of ``rows`` output rows"""
    o = out.float().reshape(-1, rows, out.shape[-1])
    return torch.stack([o.sum(1), (o * o).sum(1)], dim=-1)


@pytest.mark.parametrize("M,N,K,rpb,res", [(8192, 640, 640, 1024, True), (4096, 320, 320, 4096, True), (2048, 1280, 1280, 256, True),
                                           (1024, 640, 2560, 1024, False), (4096, 960, 320, 1024, False)])
def test_lean_linear_epilogue_statistics(env, cuda, M, N, K, rpb, res):
    """udt_gemm_desc.colstats on the lean GEMM kernels (128x128 and 128x160 tiles): per-(row slot, column) sum and sum of
    squares of the STORED bf16 values, the input of udt_gn_finalize — against torch sums of the returned output; the output
    itself must not change when statistics are requested"""
    x, wp, bp, kw, y = _linear_case(env, cuda, M, N, K, 0, res, 0, seed=21)
    kw["rows_per_batch"] = rpb
    plain = env.ops.linear(x, wp, bp, **kw)
    out = env.ops.linear(x, wp, bp, colstats=True, **kw)
    torch.cuda.synchronize()
    st = env.ops.gn_stats_of(out)
    assert st is not None, "the library declined statistics for a lean-kernel shape"
    assert torch.equal(out, plain)
    rows = rpb // st.slots_per_sample
    assert rows in (32, 64) and st.data.shape == (M // rows, N, 2)
    ref = _stats_ref(out, rows)
    err = (st.data - ref).abs()
    tol = 1e-4 * ref.abs() + 1e-3 * math.sqrt(rows)
    assert (err <= tol).all(), f"max err {err.max().item():.3e}"
    assert _rel(out, y) < REL_RMS


@pytest.mark.parametrize("B,H,W,C,N,ups,res", [(2, 64, 64, 320, 320, False, True), (3, 32, 32, 640, 640, False, False),
                                                (2, 8, 8, 1280, 1280, False, True), (2, 16, 16, 640, 640, True, False),
                                                (1, 24, 40, 128, 136, False, False)])
def test_lean_conv_epilogue_statistics_and_groupnorm_from_them(env, cuda, B, H, W, C, N, ups, res):
    """statistics out of the lean 3x3 convolution's epilogue (all three geometries), then GroupNorm(32)+SiLU built from them
    (udt_gn_finalize + udt_gn_apply_scsh) against torch group_norm of the convolution's output in fp32 and against the
    two-kernel GroupNorm (udt_gn_stats + udt_gn_apply)"""
    g = torch.Generator(device="cpu").manual_seed(22)
    x = torch.randn((B, H, W, C), generator=g).to(cuda).bfloat16()
    w = env.packing.pack_conv((torch.randn((N, C, 3, 3), generator=g) / math.sqrt(9 * C)).to(cuda))
    b = torch.randn((N,), generator=g).to(cuda)
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    r = torch.randn((B, Ho, Wo, N), generator=g).to(cuda).bfloat16() if res else None
    plain = env.ops.conv2d(x, w, b, ksize=3, upsample=ups, residual=r)
    out = env.ops.conv2d(x, w, b, ksize=3, upsample=ups, residual=r, colstats=True)
    torch.cuda.synchronize()
    st = env.ops.gn_stats_of(out)
    assert st is not None and torch.equal(out, plain)
    tot = st.data.reshape(B, st.slots_per_sample, N, 2).sum(1)
    o = out.float().reshape(B, Ho * Wo, N)
    ref = torch.stack([o.sum(1), (o * o).sum(1)], dim=-1)
    assert ((tot - ref).abs() <= 2e-4 * ref.abs() + 2e-2 * math.sqrt(Ho * Wo)).all()
    if N % 64 == 0:
        gamma = (1 + 0.1 * torch.randn((N,), generator=g)).to(cuda)
        beta = (0.1 * torch.randn((N,), generator=g)).to(cuda)
        got = env.ops.group_norm_from_stats(out, st, gamma, beta, 32, 1e-5, True)
        want = F.silu(F.group_norm(out.float().permute(0, 3, 1, 2), 32, gamma, beta, 1e-5)).permute(0, 2, 3, 1)
        two = env.ops.group_norm(out, gamma, beta, 32, 1e-5, True)
        torch.cuda.synchronize()
        assert _rel(got, want) < 6e-3, f"GroupNorm from epilogue statistics vs torch: {_rel(got, want):.3e}"
        assert (got.float() - two.float()).abs().max().item() <= 4e-2      # same statistics up to fp32 summation order


def test_groupnorm_from_statistics_two_sources(env, cuda):
    """decoder ResBlock input: GroupNorm over the channel concat of two tensors, each with its own producer statistics"""
    g = torch.Generator(device="cpu").manual_seed(23)
    B, H, C1, C2 = 2, 32, 640, 320
    mk = lambda c_in, c_out, seed: env.ops.conv2d(
        torch.randn((B, H, H, c_in), generator=g).to(cuda).bfloat16(),
        env.packing.pack_conv((torch.randn((c_out, c_in, 3, 3), generator=g) / math.sqrt(9 * c_in)).to(cuda)), None, ksize=3, colstats=True)
    a, b2 = mk(320, C1, 1), mk(320, C2, 2)
    sa, sb = env.ops.gn_stats_of(a), env.ops.gn_stats_of(b2)
    assert sa is not None and sb is not None
    gamma = (1 + 0.1 * torch.randn((C1 + C2,), generator=g)).to(cuda)
    beta = (0.1 * torch.randn((C1 + C2,), generator=g)).to(cuda)
    got = env.ops.group_norm_from_stats(a, sa, gamma, beta, 32, 1e-5, True, x2=b2, st2=sb)
    cat = torch.cat([a, b2], -1).float()
    want = F.silu(F.group_norm(cat.permute(0, 3, 1, 2), 32, gamma, beta, 1e-5)).permute(0, 2, 3, 1)
    torch.cuda.synchronize()
    assert _rel(got, want) < 6e-3


@pytest.mark.parametrize("B,H,C,N", [(4, 16, 1280, 1280), (8, 8, 1280, 1280), (3, 16, 640, 1280)])
def test_strip_groupnorm_from_epilogue_statistics(env, cuda, B, H, C, N):
    """small levels: udt_gn_strip_stats (the strip GroupNorm without its statistics pass, fed by the lean convolution's
    epilogue statistics, incl. the split-K launches of the 16x16 / 8x8 maps) vs torch group_norm and vs udt_gn_strip"""
    g = torch.Generator(device="cpu").manual_seed(24)
    x = torch.randn((B, H, H, C), generator=g).to(cuda).bfloat16()
    w = env.packing.pack_conv((torch.randn((N, C, 3, 3), generator=g) / math.sqrt(9 * C)).to(cuda))
    out = env.ops.conv2d(x, w, None, ksize=3, colstats=True)
    st = env.ops.gn_stats_of(out)
    assert st is not None and env.ops.gn_strip_ok(B, H * H, N, 0, 32)
    gamma = (1 + 0.1 * torch.randn((N,), generator=g)).to(cuda)
    beta = (0.1 * torch.randn((N,), generator=g)).to(cuda)
    got = env.ops.group_norm_from_stats(out, st, gamma, beta, 32, 1e-5, True)
    two = env.ops.group_norm(out, gamma, beta, 32, 1e-5, True)
    want = F.silu(F.group_norm(out.float().permute(0, 3, 1, 2), 32, gamma, beta, 1e-5)).permute(0, 2, 3, 1)
    torch.cuda.synchronize()
    assert _rel(got, want) < 6e-3
    assert (got.float() - two.float()).abs().max().item() <= 4e-2
