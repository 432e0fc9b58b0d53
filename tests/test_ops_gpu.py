"""Per-kernel numerics: every HIP op against a plain PyTorch fp32 reference of the same op, evaluated on
the SAME bf16-rounded inputs (so the tolerance only has to cover fp32-accumulated bf16 products and the
final bf16 rounding of the output).  Runs on the GPU box: ``pytest -m gpu``.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# tolerances (stated): outputs are bf16 (8 bit mantissa -> rel 2^-8 = 3.9e-3 per rounding)
RTOL = 1.5e-2
ATOL_BF16 = 2e-2


def _close(got, ref, atol=ATOL_BF16, rtol=RTOL, what=""):
    got = got.float()
    ref = ref.float()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = (err > tol).sum().item()
    assert bad == 0, f"{what}: {bad}/{err.numel()} off, max err {err.max().item():.4g} (ref max {ref.abs().max().item():.4g})"


def _close_signal(got, ref, rel_rms=6e-3, rel_max=3e-2, what=""):
    """signal-relative check for outputs whose magnitude shrinks with the problem size (softmax averages of N random
    values have std ~ 1/sqrt(N): at N = 4096 / 9216 the absolute tolerance of _close exceeds the signal itself).  Stated
    tolerance: error RMS <= 6e-3 of the reference RMS (bf16 output rounding alone is ~2e-3) and max |err| <= 3e-2 of
    max |ref|."""
    got = got.float()
    ref = ref.float()
    err = got - ref
    rr = (err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-30)).item()
    rm = (err.abs().max() / ref.abs().max().clamp_min(1e-30)).item()
    assert rr <= rel_rms and rm <= rel_max, f"{what}: rel RMS {rr:.3e} (<= {rel_rms:g}), max|err|/max|ref| {rm:.3e} (<= {rel_max:g})"


def _rand(shape, dev, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


@pytest.fixture(scope="module")
def ops(cuda):
    import udifftext_amd  # noqa: F401
    from udifftext_amd import lib, ops as O
    lib.load()
    assert lib.load().udt_device_arch_ok() == 1, "tests expect a gfx950 device"
    return O


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 320, 320), (1000, 640, 1280), (96, 640, 2048),
                                   (4096, 320, 2880), (512, 1280, 11520), (8, 1280, 320), (130, 4, 128)])
def test_linear_plain(ops, cuda, M, N, K):
    from udifftext_amd import packing
    x = _rand((M, K), cuda, seed=1).bfloat16()
    # asymmetric weights (transpose-detecting): scale grows with the row index
    w = _rand((N, K), cuda, 1.0 / math.sqrt(K), seed=2) * (1.0 + torch.arange(N, device=cuda)[:, None] / N)
    b = _rand((N,), cuda, seed=3)
    wp = packing.pack_linear(w)
    out = ops.linear(x, wp, packing.pad_bias(b), n_out=wp.shape[0])
    ref = x.float() @ wp[:, :K].float().t() + packing.pad_bias(b)
    _close(out, ref, what=f"linear {M}x{N}x{K}")


@pytest.mark.parametrize("n_block", [-1, 0, 3, 4, 64])
def test_linear_blocked_tile_order(ops, cuda, n_block):
    """wide outputs walk their tiles in blocks of N-tiles (decode_tile): automatic, off, forced widths that do not
    divide the 11 N-tiles (ragged last block) and one wider than the matrix — all must give the same product"""
    from udifftext_amd import lib as L, packing
    M, N, K = 520, 1408, 3072            # 3 x 11 tiles of 256x128, partial last M-tile
    x = _rand((M, K), cuda, seed=1).bfloat16()
    w = _rand((N, K), cuda, 1.0 / math.sqrt(K), seed=2) * (1.0 + torch.arange(N, device=cuda)[:, None] / N)
    wp = packing.pack_linear(w)
    lib = L.load()
    try:
        L.check(lib.udt_debug_set(b"n_block", n_block), "udt_debug_set")
        out = ops.linear(x, wp, n_out=N)
    finally:
        L.check(lib.udt_debug_set(b"n_block", -1), "udt_debug_set")
    _close(out, x.float() @ wp[:, :K].float().t(), what=f"linear blocked order n_block={n_block}")


def test_linear_epilogues(ops, cuda):
    from udifftext_amd import lib as L, packing
    M, N, K, rpb = 512, 640, 640, 128
    x = _rand((M, K), cuda, seed=1).bfloat16()
    w = packing.pack_linear(_rand((N, K), cuda, 1 / math.sqrt(K), seed=2))
    b = _rand((N,), cuda, seed=3)
    res = _rand((M, N), cuda, seed=4).bfloat16()
    rvbig = _rand((M // rpb, 3 * N), cuda, seed=5)
    rv = rvbig[:, N:2 * N]                                        # strided row vector (ld_rowvec)
    ref = x.float() @ w.float().t() + b + res.float() + rv.repeat_interleave(rpb, 0)
    out = ops.linear(x, w, b, residual=res, rowvec=rv, rows_per_batch=rpb)
    _close(out, ref, what="bias+res+rowvec")
    out32 = ops.linear(x, w, b, flags=L.GEMM_OUT_F32)
    _close(out32, x.float() @ w.float().t() + b, atol=2e-3, rtol=1e-3, what="fp32 out")
    outr = ops.linear(x, w, b, flags=L.GEMM_RELU)
    _close(outr, torch.relu(x.float() @ w.float().t() + b), what="relu")
    outs = ops.linear(x, w, b, flags=L.GEMM_SILU_OUT)
    _close(outs, F.silu(x.float() @ w.float().t() + b), what="silu")
    outa = ops.linear(x, w, None, alpha=0.125)
    _close(outa, 0.125 * (x.float() @ w.float().t()), what="alpha")


def test_linear_strided_input(ops, cuda):
    from udifftext_amd import packing
    M, K, N = 256, 320, 320
    big = _rand((M, 3 * K), cuda, seed=7).bfloat16()
    xv = big[:, K:2 * K]
    w = packing.pack_linear(_rand((N, K), cuda, 1 / math.sqrt(K), seed=8))
    out = ops.linear(xv, w)
    _close(out, xv.float() @ w.float().t(), what="strided A")


def test_geglu(ops, cuda):
    from udifftext_amd import lib as L, packing
    M, Cc = 300, 320
    inner = 4 * Cc
    x = _rand((M, Cc), cuda, seed=1).bfloat16()
    w = _rand((2 * inner, Cc), cuda, 1 / math.sqrt(Cc), seed=2)
    b = _rand((2 * inner,), cuda, 0.5, seed=3)
    wp, bp = packing.pack_geglu(w, b)
    out = ops.linear(x, wp, bp, flags=L.GEMM_GEGLU)
    assert out.shape == (M, inner)
    proj = x.float() @ w.bfloat16().float().t() + b
    val, gate = proj.chunk(2, dim=-1)
    _close(out, val * F.gelu(gate), what="geglu")


def test_transposed_out(ops, cuda):
    from udifftext_amd import lib as L, packing
    B, T, Cc = 3, 64, 320
    x = _rand((B * T, Cc), cuda, seed=1).bfloat16()
    w = packing.pack_linear(_rand((Cc, Cc), cuda, 1 / math.sqrt(Cc), seed=2))
    out = ops.linear(x, w, flags=L.GEMM_TRANSPOSED, rows_per_batch=T)
    ref = (x.float() @ w.float().t()).reshape(B, T, Cc).permute(0, 2, 1)
    _close(out, ref, what="V^T")


def test_bmm(ops, cuda):
    B, M, N, K = 2, 256, 192, 512
    a = _rand((B, M, K), cuda, seed=1).bfloat16()
    w = _rand((B, N, K), cuda, 1 / math.sqrt(K), seed=2).bfloat16()
    out = ops.bmm_nt(a, w, alpha=0.5)
    _close(out, 0.5 * torch.einsum("bmk,bnk->bmn", a.float(), w.float()), what="bmm")
    qk = _rand((B, M, 2 * K), cuda, seed=3).bfloat16()          # strided q / k views of one projection
    out = ops.bmm_nt(qk[..., :K], qk[..., K:], alpha=K ** -0.5)
    _close(out, K ** -0.5 * torch.einsum("bmk,bnk->bmn", qk[..., :K].float(), qk[..., K:].float()), what="bmm strided")


CONV_CASES = [
    # B, H, W, C1, C2, N, k, stride, pad, ups
    (2, 16, 16, 64, 0, 320, 3, 1, (1, 1), False),
    (2, 32, 32, 320, 0, 320, 3, 1, (1, 1), False),
    (1, 16, 16, 640, 640, 640, 3, 1, (1, 1), False),     # skip concat
    (2, 16, 16, 320, 0, 320, 3, 2, (1, 1), False),       # UNet downsample
    (2, 16, 16, 128, 0, 128, 3, 2, (0, 0), False),       # VAE downsample (pad right/bottom only)
    (2, 8, 8, 1280, 0, 1280, 3, 1, (1, 1), True),        # nearest x2 + conv
    (2, 8, 8, 1280, 1280, 1280, 3, 1, (1, 1), False),    # deep, split-K
    (2, 16, 16, 640, 320, 320, 1, 1, (0, 0), False),     # 1x1 skip on a concat
    (1, 24, 40, 128, 0, 4, 3, 1, (1, 1), False),         # narrow output, non-square
    # patch-staged kernel (conv3p): 32x8, 16x16 and 4x(8x8) spatial tiles, bn 128 / 160, stream-K splits
    (2, 8, 8, 1280, 0, 1280, 3, 1, (1, 1), False),       # 8x8 maps, image group only half full (B=2 of 4)
    (8, 8, 8, 640, 0, 1280, 3, 1, (1, 1), False),
    (5, 8, 8, 128, 0, 320, 3, 1, (1, 1), False),         # ragged image groups, bn 160
    (1, 64, 64, 320, 0, 320, 3, 1, (1, 1), False),
    (2, 32, 32, 640, 0, 640, 3, 1, (1, 1), False),
    (3, 16, 16, 1280, 0, 1280, 3, 1, (1, 1), False),
    (1, 32, 64, 128, 0, 128, 3, 1, (1, 1), False),       # non-square, VAE-like
    (1, 16, 32, 192, 0, 256, 3, 1, (1, 1), False),       # 3 chunks
    (2, 48, 48, 640, 0, 640, 3, 1, (1, 1), False),       # 48x48 maps (768x768 inputs): 3 x 3 tiles of 16x16
    (2, 32, 32, 640, 320, 640, 3, 1, (1, 1), False),     # patch-staged, two sources (skip concat), 15 chunks
    (4, 8, 8, 1280, 1280, 1280, 3, 1, (1, 1), False),    # 8x8 maps, two sources
    (1, 64, 64, 320, 320, 320, 3, 1, (1, 1), False),     # bn 160, two sources
    # patch-staged kernel with the nearest x2 upsampling folded in: 16x16 and 32x8 output tiles, stream-K / whole tiles
    (3, 16, 16, 640, 0, 640, 3, 1, (1, 1), True),        # -> 32x32
    (1, 32, 32, 128, 0, 256, 3, 1, (1, 1), True),        # -> 64x64
    (2, 24, 24, 192, 0, 128, 3, 1, (1, 1), True),        # -> 48x48 (16x16 tiles), 3 chunks
    (1, 16, 48, 128, 0, 128, 3, 1, (1, 1), True),        # non-square -> 32x96
    (2, 8, 8, 1280, 0, 320, 3, 1, (1, 1), True),         # N = 320 (bn 160): stays on the gather kernel
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv(ops, cuda, case):
    from udifftext_amd import packing
    B, H, W, C1, C2, N, k, stride, pad, ups = case
    x1 = _rand((B, H, W, C1), cuda, seed=1).bfloat16()
    x2 = _rand((B, H, W, C2), cuda, seed=2).bfloat16() if C2 else None
    Cin = C1 + C2
    w = _rand((N, Cin, k, k), cuda, 1 / math.sqrt(Cin * k * k), seed=3)
    w = w * (1.0 + torch.arange(k * k, device=cuda).reshape(1, 1, k, k) / 4.0)   # tap-asymmetric
    b = _rand((N,), cuda, seed=4)
    wp = packing.pack_conv(w, [C1, C2] if C2 else None)
    if stride == 2 and pad == (0, 0):
        Hv, Wv = H, W
        out_hw = ((Hv + 1 - k) // 2 + 1, (Wv + 1 - k) // 2 + 1)
    else:
        out_hw = None
    out = ops.conv2d(x1, wp, packing.pad_bias(b), ksize=k, stride=stride, pad=pad, upsample=ups, x2=x2, out_hw=out_hw,
                     n_out=wp.shape[0])
    xin = x1 if x2 is None else torch.cat([x1, x2], dim=3)
    xin = xin.float().permute(0, 3, 1, 2)
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    wq = w.bfloat16().float()
    if stride == 2 and pad == (0, 0):
        xin = F.pad(xin, (0, 1, 0, 1))
        ref = F.conv2d(xin, wq, b, stride=2, padding=0)
    else:
        ref = F.conv2d(xin, wq, b, stride=stride, padding=pad)
    ref = ref.permute(0, 2, 3, 1)
    _close(out[..., :N], ref, what=f"conv {case}")


def _colstats_ref(out_bhwc, rows, tile=None):
    """reference column statistics of a [B, H, W, N] output: per (slot of `rows` pixels in the kernel's order, column)"""
    B, H, W, N = out_bhwc.shape
    o = out_bhwc.float()
    if tile is not None:                  # patch-staged conv: pixels in (image group, tile y, tile x, in-tile) order
        TW, TH, NI = tile
        if NI == 1:
            o = o.reshape(B, H // TH, TH, W // TW, TW, N).permute(0, 1, 3, 2, 4, 5).reshape(-1, N)
        else:
            Bp = (B + NI - 1) // NI * NI
            o = torch.cat([o, torch.zeros((Bp - B, H, W, N), device=o.device)], 0).reshape(-1, N)
    else:
        o = o.reshape(-1, N)
        pad = (-o.shape[0]) % 256
        o = torch.cat([o, torch.zeros((pad, N), device=o.device)], 0)
    o = o.reshape(-1, rows, N)
    return torch.stack([o.sum(1), (o * o).sum(1)], dim=-1)            # [slots, N, 2]


@pytest.mark.parametrize("M,N,K,hw", [(2048, 640, 640, 256), (4096, 320, 320, 1024), (512, 1280, 1280, 64),
                                      (384, 1280, 2560, 64), (1000, 256, 128, 200)])
def test_linear_colstats(ops, cuda, M, N, K, hw):
    """the GEMM epilogue's per-(row slot, column) partial sums == sums of the bf16 output it stored (256x128 and 256x160
    tiles, interior and ragged tiles, residual epilogue); unsupported splits (rows_per_batch % slot != 0) give None"""
    from udifftext_amd import packing
    x = _rand((M, K), cuda, seed=1).bfloat16()
    wp = packing.pack_linear(_rand((N, K), cuda, 1.0 / math.sqrt(K), seed=2))
    b = _rand((N,), cuda, seed=3)
    res = _rand((M, N), cuda, seed=4).bfloat16()
    out = ops.linear(x, wp, b, residual=res, rows_per_batch=hw, colstats=True)
    st = ops.gn_stats_of(out)
    if hw % 32 != 0:
        assert st is None
        return
    assert st is not None
    rows = hw // st.slots_per_sample
    assert rows in (32, 64)
    ref = _colstats_ref(out.reshape(1, 1, M, N), rows)
    n = (M + rows - 1) // rows            # slots that hold rows (the 8-wave kernels pad to 256-row tiles, the lean ones to 128)
    assert st.data.shape[0] >= n
    torch.testing.assert_close(st.data[:n], ref[:n], rtol=2e-4, atol=2e-3)
    _close(out, x.float() @ wp.float().t() + b + res.float(), what="linear with colstats")


@pytest.mark.parametrize("B,H,W,C1,C2,N", [(2, 32, 32, 320, 0, 640), (1, 64, 64, 320, 0, 320), (3, 16, 16, 640, 0, 1280),
                                          (5, 8, 8, 1280, 0, 1280), (2, 32, 32, 640, 320, 640), (4, 8, 8, 1280, 1280, 1280), (1, 48, 48, 640, 640, 640),
                                          (2, 64, 64, 320, 320, 320), (2, 16, 16, 1280, 640, 1280)])
def test_gn_silu_conv3x3_fused(ops, cuda, B, H, W, C1, C2, N):
    """udt_gn_silu_conv3x3_fwd: statistics from producer epilogues -> udt_gn_finalize -> GroupNorm + SiLU applied on the
    staged patch -> conv3x3 (+ time-embedding row vector + residual) -> statistics of the output; one and two sources,
    every spatial tiling (32x8, 16x16, 4 x 8x8 with a ragged image group), both weight-tile widths.  Reference: torch
    fp32 group_norm + silu + conv2d on the same bf16 inputs."""
    from udifftext_amd import packing
    torch.manual_seed(B * 1000 + H + C1 + C2)
    # the inputs come out of "producers" (1x1 linears with colstats) so that their statistics exist
    def produce(C, seed):
        K = 128
        a = (_rand((B * H * W, K), cuda, 1.5, seed=seed) + 0.4).bfloat16()
        wq = packing.pack_linear(_rand((C, K), cuda, 1.0 / math.sqrt(K), seed=seed + 1))
        bias = _rand((C,), cuda, seed=seed + 2) * 0.5
        y = ops.linear(a, wq, bias, rows_per_batch=H * W, colstats=True)
        assert ops.gn_stats_of(y) is not None
        t = y.reshape(B, H, W, C)
        t.gn_stats = y.gn_stats
        return t
    x1 = produce(C1, 10)
    x2 = produce(C2, 20) if C2 else None
    Ct = C1 + C2
    gamma = _rand((Ct,), cuda, seed=5) * 0.2 + 1.0
    beta = _rand((Ct,), cuda, seed=6) * 0.2
    w = _rand((N, Ct, 3, 3), cuda, 1 / math.sqrt(Ct * 9), seed=3)
    w = w * (1.0 + torch.arange(9, device=cuda).reshape(1, 1, 3, 3) / 4.0)
    bias = _rand((N,), cuda, seed=4)
    temb = _rand((B, N), cuda, seed=7)
    res = _rand((B, H, W, N), cuda, seed=8).bfloat16()
    wp = packing.pack_conv(w, [C1, C2] if C2 else None)
    assert ops.conv2d(x1, wp, bias, x2=x2, probe_in_scsh=True)
    scsh = ops.gn_finalize(x1.gn_stats, C1, x2.gn_stats if C2 else None, C2, gamma, beta, B, H * W, 32, 1e-5)
    cat = x1.float() if x2 is None else torch.cat([x1, x2], dim=-1).float()
    gn = F.group_norm(cat.permute(0, 3, 1, 2), 32, gamma, beta, 1e-5)
    # the table itself: scale = rstd * gamma, shift = beta - mean * scale
    tab = scsh.reshape(B, Ct // 64, 2, 64)
    sc = tab[:, :, 0].reshape(B, Ct)
    sh = tab[:, :, 1].reshape(B, Ct)
    gn_from_table = cat * sc[:, None, None, :] + sh[:, None, None, :]
    torch.testing.assert_close(gn_from_table.permute(0, 3, 1, 2), gn, rtol=2e-3, atol=2e-3)
    out = ops.conv2d(x1, wp, bias, x2=x2, in_scsh=scsh, in_act=1, rowvec=temb, residual=res, colstats=True)
    ref = F.conv2d(F.silu(gn).bfloat16().float(), w.bfloat16().float(), bias, padding=1).permute(0, 2, 3, 1)
    ref = ref + temb[:, None, None, :] + res.float()
    _close(out, ref, atol=3e-2, what=f"gn+silu+conv3x3 fused {B,H,W,C1,C2,N}")
    # statistics of the output in the patch-staged kernel's slot order
    st = ops.gn_stats_of(out)
    assert st is not None
    rows = (H * W) // st.slots_per_sample
    tile = (32, 8, 1) if W % 32 == 0 else (16, 16, 1) if W % 16 == 0 else (8, 8, 4)
    sref = _colstats_ref(out, rows, tile)
    n = min(sref.shape[0], st.data.shape[0])
    live = B * st.slots_per_sample                               # slots of images that exist
    torch.testing.assert_close(st.data[:live], sref[:live], rtol=2e-4, atol=5e-3)
    assert n >= live
    # and the next layer's normalisation from them == group_norm of the stored output
    g2 = _rand((N,), cuda, seed=9) * 0.2 + 1.0
    b2 = _rand((N,), cuda, seed=11) * 0.2
    tab2 = ops.gn_finalize(st, N, None, 0, g2, b2, B, H * W, 32, 1e-5).reshape(B, N // 64, 2, 64)
    got2 = out.float() * tab2[:, :, 0].reshape(B, 1, 1, N) + tab2[:, :, 1].reshape(B, 1, 1, N)
    ref2 = F.group_norm(out.float().permute(0, 3, 1, 2), 32, g2, b2, 1e-5).permute(0, 2, 3, 1)
    torch.testing.assert_close(got2, ref2, rtol=2e-3, atol=3e-3)


def test_fused_gn_matches_unfused_resblock(ops, cuda):
    """hipnn.Conv2d(norm=...) takes the fused path when statistics ride on the input and the unfused GroupNorm kernels
    otherwise; both must agree to bf16 rounding of the normalised activations"""
    import sgm.modules.hipnn as H
    from sgm.util import skip_param_init
    from udifftext_amd import packing, synth
    with skip_param_init():
        norm = H.GroupNorm(32, 640)
        conv = H.Conv2d(640, 320, 3, padding=1)
    synth.fill_module_(norm, "t.norm."); synth.fill_module_(conv, "t.conv.")
    norm.to(cuda); conv.to(cuda)
    a = (_rand((2 * 32 * 32, 128), cuda, 1.5, seed=1) + 0.3).bfloat16()
    wq = packing.pack_linear(_rand((640, 128), cuda, 0.1, seed=2))
    y = ops.linear(a, wq, None, rows_per_batch=1024, colstats=True)
    x = H.carry_stats(y.reshape(2, 32, 32, 640), y)
    prev = (H.FUSE_GN, H.GN_EPI)
    try:
        H.FUSE_GN, H.GN_EPI = True, False
        fused = conv(x, norm=norm, norm_silu=True, colstats=True)
        assert ops.gn_stats_of(fused) is not None
        H.FUSE_GN, H.GN_EPI = False, False
        plain = conv(x, norm=norm, norm_silu=True, colstats=True)
        assert ops.gn_stats_of(plain) is None
        # the default since round 3: statistics from the producer's epilogue, finalize + streaming apply, unfused convolution
        H.FUSE_GN, H.GN_EPI = False, True
        ops.WORK_COUNTER = {}
        epi = conv(x, norm=norm, norm_silu=True, colstats=True)
        used = ops.WORK_COUNTER.get("gn_from_epilogue_stats", 0)
        ops.WORK_COUNTER = None
        assert used == 1 and ops.gn_stats_of(epi) is not None
    finally:
        H.FUSE_GN, H.GN_EPI = prev
    _close(fused, plain, atol=3e-2, what="fused vs unfused GN+SiLU+conv")
    assert (fused.float() - plain.float()).abs().mean().item() < 4e-3
    _close(epi, plain, atol=3e-2, what="GroupNorm from epilogue statistics vs the gn_stats / gn_apply pair")
    assert (epi.float() - plain.float()).abs().mean().item() < 4e-3


def test_conv_epilogue(ops, cuda):
    from udifftext_amd import packing
    B, H, W, Cc, N = 2, 16, 16, 320, 320
    x = _rand((B, H, W, Cc), cuda, seed=1).bfloat16()
    w = _rand((N, Cc, 3, 3), cuda, 1 / math.sqrt(Cc * 9), seed=3)
    b = _rand((N,), cuda, seed=4)
    res = _rand((B, H, W, N), cuda, seed=5).bfloat16()
    temb = _rand((B, N), cuda, seed=6)
    out = ops.conv2d(x, packing.pack_conv(w), b, residual=res, rowvec=temb)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.bfloat16().float(), b, padding=1).permute(0, 2, 3, 1)
    ref = ref + res.float() + temb[:, None, None, :]
    _close(out, ref, what="conv epilogue")


@pytest.mark.parametrize("B,H,N,Nk", [(2, 5, 256, 256), (1, 10, 1024, 1024), (2, 20, 64, 64), (1, 20, 16, 16),
                                      (1, 5, 4096, 4096), (1, 5, 200, 136), (1, 5, 9216, 9216), (2, 3, 1000, 72), (1, 2, 72, 1096)])
def test_flash_attention(ops, cuda, B, H, N, Nk):
    Cc = H * 64
    qkv = _rand((B, max(N, Nk), 3 * Cc), cuda, seed=1).bfloat16()
    q = qkv[:, :N, :Cc]
    k = qkv[:, :Nk, Cc:2 * Cc]
    v = qkv[:, :Nk, 2 * Cc:]
    vt = v.permute(0, 2, 1).contiguous()
    out = ops.attention(q, k, vt, H, 0.125)
    qh = q.float().reshape(B, N, H, 64).permute(0, 2, 1, 3)
    kh = k.float().reshape(B, Nk, H, 64).permute(0, 2, 1, 3)
    vh = v.float().reshape(B, Nk, H, 64).permute(0, 2, 1, 3)
    ref = F.scaled_dot_product_attention(qh, kh, vh).permute(0, 2, 1, 3).reshape(B, N, Cc)
    _close(out, ref, what=f"attn {B,H,N,Nk}")
    _close_signal(out, ref, what=f"attn {B,H,N,Nk}")
    # V row-major (column range of the same q|k|v rows): LDS transpose reads instead of a transposed V
    out2 = ops.attention_rowv(q, k, v, H, 0.125)
    _close(out2, ref, what=f"attn rowv {B,H,N,Nk}")
    _close_signal(out2, ref, what=f"attn rowv {B,H,N,Nk}")
    # (the row-major-V kernel defers the online-softmax rescale, so the two layouts agree to rounding, not bit for bit)
    _close(out2, out, what=f"attn rowv vs V^T {B,H,N,Nk}")


@pytest.mark.parametrize("B,H,N,Nk", [(2, 20, 256, 256), (1, 5, 200, 136), (1, 3, 72, 1096), (3, 2, 130, 40), (2, 10, 1024, 1024),
                                      (1, 20, 4, 4), (2, 5, 12, 12), (1, 10, 36, 36)])      # (round 6: key counts that are not multiples of 8 —
#                                       the 2 x 2 middle block of a 128 x 128 image attends over 4 tokens)
def test_flash_attention_row_major_v_ragged_shapes(ops, cuda, B, H, N, Nk):
    """udt_attn_rowv_fwd (the UNet's self-attention kernel) on ragged query / key counts on both sides of a 128-query block / a
    64-key tile boundary, several samples and heads: signal-relative tolerance against torch SDPA in fp32, bit-reproducible"""
    Cc = H * 64
    qkv = _rand((B, max(N, Nk), 3 * Cc), cuda, seed=5).bfloat16()
    q, k, v = qkv[:, :N, :Cc], qkv[:, :Nk, Cc:2 * Cc], qkv[:, :Nk, 2 * Cc:]
    qh = q.float().reshape(B, N, H, 64).permute(0, 2, 1, 3)
    kh = k.float().reshape(B, Nk, H, 64).permute(0, 2, 1, 3)
    vh = v.float().reshape(B, Nk, H, 64).permute(0, 2, 1, 3)
    ref = F.scaled_dot_product_attention(qh, kh, vh).permute(0, 2, 1, 3).reshape(B, N, Cc)
    out = ops.attention_rowv(q, k, v, H, 0.125)
    again = ops.attention_rowv(q, k, v, H, 0.125)
    torch.cuda.synchronize()
    _close(out, ref, what=f"attn rowv {B,H,N,Nk}")
    _close_signal(out, ref, what=f"attn rowv {B,H,N,Nk}")
    assert torch.equal(out, again)


def test_flash_attention_spike(ops, cuda):
    """a key row that dominates one query late in the sequence (forces a large online-softmax rescale)"""
    B, H, N = 1, 5, 512
    Cc = H * 64
    q = _rand((B, N, Cc), cuda, seed=1).bfloat16()
    k = _rand((B, N, Cc), cuda, seed=2).bfloat16()
    v = _rand((B, N, Cc), cuda, seed=3).bfloat16()
    k[0, 400] = (q[0, 17].float() * 4).bfloat16()
    out = ops.attention(q, k, v.permute(0, 2, 1).contiguous(), H, 0.125)
    qh, kh, vh = (t.float().reshape(B, N, H, 64).permute(0, 2, 1, 3) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qh, kh, vh).permute(0, 2, 1, 3).reshape(B, N, Cc)
    _close(out, ref, what="attn spike")
    _close_signal(out, ref, what="attn spike")
    # the row-major-V kernel skips the rescale while the maximum grows by < 2^8: the spike must take the rescale branch, and
    # a slowly growing maximum (every key a little larger than the last) must stay exact without it
    o_sp = ops.attention_rowv(q, k, v, H, 0.125)
    _close(o_sp, ref, what="attn rowv spike")
    _close_signal(o_sp, ref, what="attn rowv spike")
    # the spiked query row itself (its softmax is ~one-hot on key 400: the output must be v[400])
    _close_signal(o_sp[0, 17], ref[0, 17], what="attn rowv spike, the spiked row")
    k2 = (q[0, 17].float()[None, :] * torch.linspace(0.0, 1.5, N, device=cuda)[:, None]).bfloat16()[None]
    kh2 = k2.float().reshape(B, N, H, 64).permute(0, 2, 1, 3)
    ref2 = F.scaled_dot_product_attention(qh, kh2, vh).permute(0, 2, 1, 3).reshape(B, N, Cc)
    o_gr = ops.attention_rowv(q, k2.contiguous(), v, H, 0.125)
    _close(o_gr, ref2, what="attn rowv slowly growing maximum")
    _close_signal(o_gr, ref2, what="attn rowv slowly growing maximum")


@pytest.mark.parametrize("B,N,Nk", [(2, 1024, 1024), (1, 4096, 4096), (3, 100, 77), (1, 33, 2050), (1, 9216, 9216), (4, 64, 64)])
def test_flash_attention_d512(ops, cuda, B, N, Nk):
    """udt_attn512_fwd (the VAE mid-block attention: one head of 512 dims, reference model.py:236-260) on the column ranges of
    one q|k|v projection vs torch scaled_dot_product_attention in fp32; N = 9216 is the 768 x 768 decode (96 x 96 latents),
    the ragged cases end inside a 32-query block / a 32-key tile"""
    qkv = _rand((B, max(N, Nk), 3 * 512), cuda, seed=1).bfloat16()
    q, k, v = qkv[:, :N, :512], qkv[:, :Nk, 512:1024], qkv[:, :Nk, 1024:]
    out = ops.attention_d512(q, k, v, 512 ** -0.5)
    ref = F.scaled_dot_product_attention(q.float(), k.float(), v.float())
    _close(out, ref, what=f"attn512 {B,N,Nk}")
    _close_signal(out, ref, what=f"attn512 {B,N,Nk}")
    # the same launch twice: bit-identical (the four waves sum the partial score tiles in a fixed order)
    assert torch.equal(out, ops.attention_d512(q, k, v, 512 ** -0.5))
    # the key-split form (planned by the library for the small grids: (1, 4096, 4096), (1, 33, 2050), (2, 1024, 1024) ...) and the
    # one-workgroup-per-query-tile form agree to the rounding of the output
    whole = ops.attention_d512(q, k, v, 512 ** -0.5, key_split=False)
    _close_signal(whole, ref, what=f"attn512 without key split {B,N,Nk}")
    _close_signal(out, whole.float(), rel_rms=8e-3, what=f"attn512 key split vs whole {B,N,Nk}")


def test_flash_attention_d512_key_split_contract(ops, cuda):
    """udt_attn512_workspace_bytes / udt_attn512_split_fwd: the plan (split only when the query tiles leave CUs idle and there are
    >= 512 keys), a short workspace is refused, NULL workspace = the unsplit kernel, and a spike in the LAST key slice survives the
    merge (its slice's maximum dominates the others)"""
    import ctypes as C
    from udifftext_amd import lib as L
    lib = L.load()
    assert lib.udt_attn512_workspace_bytes(4, 4096, 4096) == 0            # 256 query tiles: no split
    assert lib.udt_attn512_workspace_bytes(1, 4096, 4096) == 1 * 4 * 4096 * 514 * 4
    assert lib.udt_attn512_workspace_bytes(1, 64, 4096) == 1 * 7 * 64 * 514 * 4
    assert lib.udt_attn512_workspace_bytes(1, 9216, 9216) == 1 * 5 * 9216 * 514 * 4        # 144 query tiles: three rounds of a fifth
    assert lib.udt_attn512_workspace_bytes(1, 64, 300) == 0               # fewer than two slices of 8 tiles
    assert lib.udt_attn512_workspace_bytes(0, 64, 4096) == 0
    B, N = 1, 2048
    q = _rand((B, N, 512), cuda, seed=4).bfloat16()
    k = _rand((B, N, 512), cuda, seed=5).bfloat16()
    v = _rand((B, N, 512), cuda, seed=6).bfloat16()
    k[0, N - 3] = (q[0, 5].float() * 3).bfloat16()
    ref = F.scaled_dot_product_attention(q.float(), k.float(), v.float())
    out = ops.attention_d512(q, k, v, 512 ** -0.5)
    _close_signal(out, ref, what="attn512 key split, spike in the last slice")
    _close_signal(out[0, 5], ref[0, 5], what="attn512 key split, the spiked row")
    need = lib.udt_attn512_workspace_bytes(B, N, N)
    assert need > 0
    o2 = torch.empty_like(out)
    small = torch.empty((need - 16,), dtype=torch.uint8, device=cuda)
    args = (q.data_ptr(), k.data_ptr(), v.data_ptr(), o2.data_ptr(), B, N, N, 512, 512, 512, 512, N * 512, N * 512, N * 512, N * 512, 512 ** -0.5)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.udt_attn512_split_fwd(*args, small.data_ptr(), need - 16, st) == -3          # UDT_ERR_WORKSPACE
    assert lib.udt_attn512_split_fwd(*args, None, 0, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(o2, ops.attention_d512(q, k, v, 512 ** -0.5, key_split=False))


def test_flash_attention_d512_spike_and_block_form(ops, cuda):
    """a late key that dominates one query (large online-softmax rescale), and agreement with the query-block
    GEMM -> softmax -> GEMM form the VAE block used before (UDT_ATTN512=0)"""
    B, N = 1, 640
    q = _rand((B, N, 512), cuda, seed=1).bfloat16()
    k = _rand((B, N, 512), cuda, seed=2).bfloat16()
    v = _rand((B, N, 512), cuda, seed=3).bfloat16()
    k[0, 500] = (q[0, 17].float() * 3).bfloat16()
    out = ops.attention_d512(q, k, v, 512 ** -0.5)
    ref = F.scaled_dot_product_attention(q.float(), k.float(), v.float())
    _close(out, ref, what="attn512 spike")
    _close_signal(out, ref, what="attn512 spike")
    _close_signal(out[0, 17], ref[0, 17], what="attn512 spike, the spiked row")
    s = ops.bmm_nt(q, k, alpha=512 ** -0.5)
    ops.softmax_rows_(s)
    blk = ops.bmm_nt(s, v.permute(0, 2, 1).contiguous())
    _close(out, blk.float(), what="attn512 vs block form")
    _close_signal(out, blk.float(), rel_rms=8e-3, what="attn512 vs block form (both bf16-rounded)")


@pytest.mark.parametrize("B,H,D,N,Lc", [(2, 5, 64, 1024, 12), (4, 20, 64, 64, 12), (3, 8, 256, 12, 12), (1, 10, 64, 300, 1)])
def test_xattention(ops, cuda, B, H, D, N, Lc):
    Cc = H * D
    q = _rand((B, N, Cc), cuda, seed=1).bfloat16()
    kv = _rand((B, Lc, 2 * Cc), cuda, seed=2).bfloat16()
    k, v = kv[..., :Cc], kv[..., Cc:]
    probs = torch.empty((B * H, N, Lc), dtype=torch.float32, device=cuda)
    out = ops.xattention(q, k, v, H, D, D ** -0.5, probs=probs)
    qh = q.float().reshape(B, N, H, D).permute(0, 2, 1, 3)
    kh = k.float().reshape(B, Lc, H, D).permute(0, 2, 1, 3)
    vh = v.float().reshape(B, Lc, H, D).permute(0, 2, 1, 3)
    sim = qh @ kh.transpose(-1, -2) * D ** -0.5
    sim = torch.softmax(sim, dim=-1) if Lc > 1 else torch.sigmoid(sim)   # attention.py:159-162
    ref = (sim @ vh).permute(0, 2, 1, 3).reshape(B, N, Cc)
    _close(out, ref, what="xattn out")
    _close(probs, sim.reshape(B * H, N, Lc), atol=1e-4, rtol=1e-3, what="xattn probs")


@pytest.mark.parametrize("B,N,heads,Lc,zero", [(3, 4096, 5, 12, 1), (2, 1024, 10, 12, 0), (4, 64, 20, 12, 2), (2, 256, 20, 9, 2)])
def test_tattn_fused(ops, cuda, B, N, heads, Lc, zero):
    """udt_tattn_prepare + udt_tattn_fused (LayerNorm, to_q, 12-token softmax attention, to_out, bias, residual in one
    launch on folded tables) against the reference op sequence in torch fp32 (attention.py:140-174,326-333) on the same
    bf16 inputs / weights; the first `zero` samples attend to an all-zero context (x + bias)"""
    from udifftext_amd import packing
    C = heads * 64
    Dc = 256
    x = (_rand((B, N, C), cuda, 1.5, seed=1) + 0.3).bfloat16()
    ctx = _rand((B, Lc, Dc), cuda, seed=2).bfloat16()
    ctx[:zero] = 0
    g = _rand((C,), cuda, seed=3) * 0.2 + 1.0
    be = _rand((C,), cuda, seed=4) * 0.2
    wq = _rand((C, C), cuda, 1 / math.sqrt(C), seed=5)
    wk = _rand((C, Dc), cuda, 1 / math.sqrt(Dc), seed=6)
    wv = _rand((C, Dc), cuda, 1 / math.sqrt(Dc), seed=7)
    wo = _rand((C, C), cuda, 1 / math.sqrt(C), seed=8)
    bo = _rand((C,), cuda, seed=9) * 0.3
    kvw = packing.pack_linear(torch.cat([wk, wv], 0))
    kv = ops.linear(ctx.reshape(B * Lc, Dc), kvw).reshape(B, Lc, 2 * C)
    tabs = ops.tattn_prepare(kv, packing.pack_linear(wq), packing.pack_linear(wo), g, be, heads, 64 ** -0.5)
    out = ops.tattn_fused(x, tabs, bo, heads, zero, 1e-5)
    # reference
    xn = F.layer_norm(x.float(), (C,), g, be, 1e-5)
    q = xn @ wq.bfloat16().float().t()
    k, v = kv[..., :C].float(), kv[..., C:].float()
    qh = q.reshape(B, N, heads, 64).permute(0, 2, 1, 3)
    kh = k.reshape(B, Lc, heads, 64).permute(0, 2, 1, 3)
    vh = v.reshape(B, Lc, heads, 64).permute(0, 2, 1, 3)
    sim = (qh @ kh.transpose(-1, -2)) * 64 ** -0.5
    pr = sim.softmax(-1)           # (a single-token context takes the reference's sigmoid branch: never folded, see attention.py)
    o = (pr @ vh).permute(0, 2, 1, 3).reshape(B, N, C)
    ref = o @ wo.bfloat16().float().t() + bo + x.float()
    ref[:zero] = x[:zero].float() + bo
    _close(out, ref, atol=3e-2, what=f"fused t_attn {B,N,heads,Lc,zero}")
    # zero-context rows are bit-exact x + bias
    assert torch.equal(out[:zero], (x[:zero].float() + bo).bfloat16())


def test_softmax_rows(ops, cuda):
    x = _rand((300, 1024), cuda, 3.0, seed=1).bfloat16()
    ref = torch.softmax(x.float(), dim=-1)
    ops.softmax_rows_(x)
    _close(x, ref, atol=1e-3, what="softmax rows")


@pytest.mark.parametrize("B,HW,Cc,silu,eps", [(2, 4096, 320, True, 1e-5), (2, 1024, 960, True, 1e-5),
                                              (3, 64, 2560, True, 1e-5), (1, 16384, 128, True, 1e-6),
                                              (2, 256, 1280, False, 1e-6), (1, 100, 512, True, 1e-6)])
def test_group_norm(ops, cuda, B, HW, Cc, silu, eps):
    x = (_rand((B, HW, Cc), cuda, 2.0, seed=1) + 0.7).bfloat16()
    g = _rand((Cc,), cuda, seed=2) * 0.2 + 1.0
    b = _rand((Cc,), cuda, seed=3) * 0.2
    out = ops.group_norm(x, g, b, 32, eps, silu)
    ref = F.group_norm(x.float().permute(0, 2, 1), 32, g, b, eps)
    if silu:
        ref = F.silu(ref)
    _close(out, ref.permute(0, 2, 1), what=f"gn {B,HW,Cc}")


@pytest.mark.parametrize("C1,C2", [(1280, 640), (640, 320), (1280, 1280)])
def test_group_norm_concat(ops, cuda, C1, C2):
    B, HW = 2, 256
    x1 = (_rand((B, HW, C1), cuda, 2.0, seed=1) + 0.7).bfloat16()
    x2 = (_rand((B, HW, C2), cuda, 0.5, seed=4) - 0.3).bfloat16()
    g = _rand((C1 + C2,), cuda, seed=2) * 0.2 + 1.0
    b = _rand((C1 + C2,), cuda, seed=3) * 0.2
    out = ops.group_norm(x1, g, b, 32, 1e-5, True, x2=x2)
    cat = torch.cat([x1, x2], dim=-1).float()
    ref = F.silu(F.group_norm(cat.permute(0, 2, 1), 32, g, b, 1e-5)).permute(0, 2, 1)
    _close(out, ref, what=f"gn concat {C1}+{C2}")


@pytest.mark.parametrize("B,HW,C1,C2", [(8, 256, 1280, 0), (8, 64, 1280, 1280), (4, 256, 1280, 640), (3, 64, 1280, 640), (2, 576, 640, 0),
                                        (1, 64, 512, 0), (5, 256, 320, 320)])
def test_group_norm_strip_matches_two_kernel_path(ops, cuda, B, HW, C1, C2):
    """the one-launch strip kernel (udt_gn_strip) against udt_gn_stats + udt_gn_apply and the torch reference"""
    import udifftext_amd.ops as O
    x1 = (_rand((B, HW, C1), cuda, 2.0, seed=1) + 0.7).bfloat16()
    x2 = (_rand((B, HW, C2), cuda, 0.5, seed=4) - 0.3).bfloat16() if C2 else None
    g = _rand((C1 + C2,), cuda, seed=2) * 0.2 + 1.0
    b = _rand((C1 + C2,), cuda, seed=3) * 0.2
    assert O.L.load().udt_gn_strip_ok(B, HW, C1, C2, 32) == 1
    prev = O.GN_STRIP
    try:
        O.GN_STRIP = False
        two = ops.group_norm(x1, g, b, 32, 1e-5, True, x2=x2)
        O.GN_STRIP = True
        one = ops.group_norm(x1, g, b, 32, 1e-5, True, x2=x2)
    finally:
        O.GN_STRIP = prev
    assert (one.float() - two.float()).abs().max().item() <= 2e-2 and (one != two).float().mean().item() < 0.02
    cat = x1.float() if x2 is None else torch.cat([x1, x2], dim=-1).float()
    ref = F.silu(F.group_norm(cat.permute(0, 2, 1), 32, g, b, 1e-5)).permute(0, 2, 1)
    _close(one, ref, what=f"gn strip {B,HW,C1,C2}")


@pytest.mark.parametrize("rows,Cc", [(1000, 320), (513, 640), (64, 1280), (36, 2048)])
def test_layer_norm(ops, cuda, rows, Cc):
    x = (_rand((rows, Cc), cuda, 2.0, seed=1) + 0.3).bfloat16()
    g = _rand((Cc,), cuda, seed=2) * 0.2 + 1.0
    b = _rand((Cc,), cuda, seed=3) * 0.2
    out = ops.layer_norm(x, g, b, 1e-5)
    _close(out, F.layer_norm(x.float(), (Cc,), g, b, 1e-5), what=f"ln {rows,Cc}")


def test_sampler_elementwise(ops, cuda):
    B, h, w = 3, 16, 24
    x = _rand((B, 4, h, w), cuda, 10.0, seed=1)
    xin = torch.zeros((2 * B, h, w, 64), dtype=torch.bfloat16, device=cuda)
    xin[..., 4:9] = 1.5
    ops.unet_input(x, xin, 0.37)
    ref = (x * 0.37).permute(0, 2, 3, 1)
    _close(xin[:B, ..., :4], ref, what="unet_input uc half")
    _close(xin[B:, ..., :4], ref, what="unet_input c half")
    assert (xin[..., 4:9].float() == 1.5).all() and (xin[..., 9:].float() == 0).all()

    eps = _rand((2 * B, h, w, 4), cuda, seed=2)
    sigma, sigma_next, scale = 3.2, 2.9, 5.0
    x0 = x.clone()
    den = torch.empty_like(x)
    ops.cfg_euler_step(x, eps, sigma, sigma_next, scale, denoised=den, c_out=-sigma)
    eu, ec = eps[:B].permute(0, 3, 1, 2), eps[B:].permute(0, 3, 1, 2)
    du, dc = x0 - sigma * eu, x0 - sigma * ec
    dref = du + scale * (dc - du)
    xref = x0 + (x0 - dref) / sigma * (sigma_next - sigma)
    assert torch.allclose(den, dref, atol=1e-4, rtol=1e-5)
    assert torch.allclose(x, xref, atol=1e-4, rtol=1e-5)


def test_layout_and_misc(ops, cuda):
    x = _rand((2, 9, 8, 12), cuda, seed=1)
    y = ops.nchw_to_nhwc(x, 64, 0.5)
    _close(y[..., :9], (x * 0.5).permute(0, 2, 3, 1), what="nchw->nhwc")
    assert (y[..., 9:].float() == 0).all()
    back = ops.nhwc_to_nchw(y, 9)
    _close(back, x * 0.5, what="nhwc->nchw")
    f32 = _rand((2, 8, 12, 8), cuda, seed=2)
    assert torch.equal(ops.nhwc_to_nchw(f32, 4), f32[..., :4].permute(0, 3, 1, 2).contiguous())

    dst = torch.zeros((2, 8, 12, 64), dtype=torch.bfloat16, device=cuda)
    src = _rand((2, 5, 8, 12), cuda, seed=3)
    ops.nhwc_set_channels(src, dst, 4)
    _close(dst[..., 4:9], src.permute(0, 2, 3, 1), what="set channels")

    mom = _rand((2, 8, 12, 8), cuda, seed=4)
    mom[..., 4:] *= 20
    noise = _rand((2, 4, 8, 12), cuda, seed=5)
    z = ops.posterior_sample(mom, noise, 0.18215)
    mean = mom[..., :4].permute(0, 3, 1, 2)
    logvar = mom[..., 4:].permute(0, 3, 1, 2).clamp(-30, 20)
    assert torch.allclose(z, 0.18215 * (mean + torch.exp(0.5 * logvar) * noise), rtol=1e-5, atol=1e-5)

    t = torch.tensor([999.0, 979.0, 19.0, 0.0], device=cuda)
    te = ops.timestep_embedding(t, 320)
    freqs = torch.exp(-math.log(10000) * torch.arange(160, device=cuda, dtype=torch.float32) / 160)
    args = t[:, None] * freqs[None]
    _close(te, torch.cat([torch.cos(args), torch.sin(args)], -1), atol=1e-2, what="timestep embedding")

    mask = (_rand((2, 1, 64, 96), cuda, seed=6) > 0).float()
    md = ops.mask_downsample(mask)
    assert torch.allclose(md, F.interpolate(mask, scale_factor=0.125, mode="bilinear"), atol=1e-6)

    idx = torch.randint(0, 95, (24,), device=cuda, dtype=torch.int32)
    table = _rand((95, 256), cuda, seed=7)
    pe = _rand((12, 256), cuda, seed=8)
    emb = ops.embed_tokens(idx, table, pe)
    _close(emb, table[idx.long()] + pe.repeat(2, 1), what="embed")

    a = _rand((1024,), cuda, seed=9).bfloat16()
    b = _rand((1024,), cuda, seed=10).bfloat16()
    ref = a.float() + b.float()
    _close(ops.add_(a, b), ref, what="add")


@pytest.mark.parametrize("size,Hm", [(16, 64), (64, 512), (96, 768)])
def test_local_loss(ops, cuda, size, Hm):
    """96x96 maps are what the 768x768 noise search scores (t_attn at the first UNet level)"""
    B, heads, Lc, seg_l = 2, 5, 12, 12
    n = size * size
    probs = torch.softmax(_rand((B * heads, n, Lc), cuda, 2.0, seed=1), dim=-1).contiguous()
    mask = torch.zeros((B, 1, Hm, Hm), device=cuda)
    mask[:, :, (5 * Hm) // 16:(5 * Hm) // 8, Hm // 8:(7 * Hm) // 8] = 1
    seg = torch.zeros((B, seg_l), device=cuda)
    seg[0, :4] = 1
    seg[1, :9] = 1
    xs = torch.arange(3, device=cuda).float()
    g1 = torch.exp(-(xs - 1) ** 2 / 2)
    gk = (g1[:, None] * g1[None, :])
    gk = (gk / gk.sum()).reshape(9).contiguous()
    loss = torch.zeros((B,), device=cuda)
    ops.local_loss_accumulate(probs, mask, seg, gk, loss, heads, size)
    am = probs.reshape(B, heads, n, Lc)[..., :seg_l].permute(0, 1, 3, 2).mean(1).reshape(B, seg_l, size, size)
    am = F.conv2d(am, gk.reshape(1, 1, 3, 3).repeat(seg_l, 1, 1, 1), padding=1, groups=seg_l).reshape(B, seg_l, n)
    mm = F.interpolate(mask, (size, size)).tile((1, seg_l, 1, 1)).reshape(B, seg_l, n)
    pl = (mm * am).max(-1)[0] + (1 - seg)
    ref = -pl.min(-1)[0]
    assert torch.allclose(loss, ref, atol=1e-5, rtol=1e-4), (loss, ref)


def test_async_error_word_and_launch_context(ops, cuda):
    """the stream-K kernels report a partner time-out through the workspace's error word: the host check must see it,
    raise, and leave the workspace usable; cu_share travels in the descriptor (no process-global state)"""
    import math
    from udifftext_amd import lib as L, packing
    ws = ops.Workspace(cuda)
    ws.check()                                                   # clean workspace: no error
    ws.buf[4 * 1023:4 * 1024].view(torch.int32).fill_(1)         # what a timed-out finisher stores
    with pytest.raises(L.UdtError, match="timed out"):
        ws.check()
    ws.check()                                                   # header re-zeroed by the failed check
    assert int(ws.buf[:4096].view(torch.int32).abs().sum()) == 0
    # a deep-K GEMM (stream-K plan: slabs + flags) on an owned workspace, whole device and half the CUs
    M, N, K = 512, 1280, 11520
    x = _rand((M, K), cuda, seed=1).bfloat16()
    wp = packing.pack_linear(_rand((N, K), cuda, 1.0 / math.sqrt(K), seed=2))
    ref = x.float() @ wp.float().t()
    for share in (1, 2, 4):
        with ops.launch_context(cu_share=share, workspace=ws):
            d = ops.gemm_desc()
            assert d.cu_share == share
            out = ops.linear(x, wp)
        ws.check()
        _close(out, ref, what=f"stream-K linear cu_share={share}")
    assert ops.gemm_desc().cu_share == 1                          # context restored
    small = ops.Workspace(cuda, 8192)
    with pytest.raises(L.UdtError, match="too small"), ops.launch_context(workspace=small):
        ops.linear(x, wp)
    # wrong-result measurement switches are not part of the product library
    assert L.load().udt_debug_set(b"no_epi", 1) != 0 and L.load().udt_debug_set(b"cu_share", 2) != 0


def test_error_codes(ops, cuda):
    from udifftext_amd import packing
    x = torch.zeros((128, 100), dtype=torch.bfloat16, device=cuda)      # K not a multiple of 64
    w = torch.zeros((64, 100), dtype=torch.bfloat16, device=cuda)
    with pytest.raises(ValueError):
        ops.linear(x, w)
    with pytest.raises(ValueError):
        ops.attention(torch.zeros((1, 12, 64), dtype=torch.bfloat16, device=cuda),
                      torch.zeros((1, 12, 64), dtype=torch.bfloat16, device=cuda),
                      torch.zeros((1, 64, 12), dtype=torch.bfloat16, device=cuda), 1, 0.125)   # nk % 8


def _packed_tensor(lib, h, which, shape, dtype):
    """copy of a packed-weight handle's buffer as a torch tensor (the handle owns its memory: hipMemcpy device to device)"""
    import ctypes as C
    ptr = {"w": lib.udt_packed_weight, "b": lib.udt_packed_bias, "s": lib.udt_packed_colscale}[which](h)
    assert ptr
    out = torch.empty(shape, dtype=dtype, device="cuda")
    torch.cuda.synchronize()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(out.data_ptr(), ptr, out.numel() * out.element_size(), 3) == 0      # 3 = hipMemcpyDeviceToDevice
    return out


def test_pack_handles_match_python_packing(ops, cuda):
    """udt_pack_linear / udt_pack_conv (+ udt_free_packed): the C-ABI packers give bit-identical buffers to
    udifftext_amd/packing.py (bf16 linear with ragged N / K, GEGLU row permutation + bias, e4m3 + per-channel scales, a 3x3
    convolution over two concatenated sources padded to 64 separately), and a GEMM runs straight off a handle"""
    import ctypes as C
    from udifftext_amd import lib as L, packing
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream

    def pack_linear(w, b, dtype, geglu):
        h = C.c_void_p()
        L.check(lib.udt_pack_linear(w.data_ptr(), b.data_ptr() if b is not None else None, w.shape[0], w.shape[1], dtype, geglu,
                                    C.byref(h), st), "udt_pack_linear")
        return h

    w = _rand((322, 200), cuda, seed=1); b = _rand((322,), cuda, seed=2)
    h = pack_linear(w, b, L.DTYPE_BF16, 0)
    ref = packing.pack_linear(w)
    assert (lib.udt_packed_dim(h, 1), lib.udt_packed_dim(h, 3)) == tuple(ref.shape) == (324, 256)
    assert torch.equal(_packed_tensor(lib, h, "w", ref.shape, torch.bfloat16).view(torch.int16), ref.view(torch.int16))
    assert torch.equal(_packed_tensor(lib, h, "b", (324,), torch.float32), packing.pad_bias(b))
    # a GEMM straight off the handle's pointers
    x = _rand((256, 200), cuda, seed=3).bfloat16()
    xp = torch.zeros((256, 256), dtype=torch.bfloat16, device=cuda); xp[:, :200] = x
    got = ops.linear(xp, _packed_tensor(lib, h, "w", ref.shape, torch.bfloat16), _packed_tensor(lib, h, "b", (324,), torch.float32))
    _close(got[:, :322], x.float() @ w.bfloat16().float().t() + b, what="GEMM off a packed handle")
    L.check(lib.udt_free_packed(h), "udt_free_packed")

    wg = _rand((2 * 640, 320), cuda, seed=4); bg = _rand((2 * 640,), cuda, seed=5)
    h = pack_linear(wg, bg, L.DTYPE_BF16, 1)
    rw, rb = packing.pack_geglu(wg, bg)
    assert torch.equal(_packed_tensor(lib, h, "w", rw.shape, torch.bfloat16).view(torch.int16), rw.view(torch.int16))
    assert torch.equal(_packed_tensor(lib, h, "b", rb.shape, torch.float32), rb)
    L.check(lib.udt_free_packed(h), "udt_free_packed")

    h = pack_linear(w, None, L.DTYPE_FP8_E4M3, 0)
    rq, rs = packing.pack_linear_fp8(w)
    assert torch.equal(_packed_tensor(lib, h, "s", rs.shape, torch.float32), rs)
    assert torch.equal(_packed_tensor(lib, h, "w", rq.shape, torch.uint8), rq)
    L.check(lib.udt_free_packed(h), "udt_free_packed")

    wc = _rand((130, 96 + 40, 3, 3), cuda, seed=6); bc = _rand((130,), cuda, seed=7)
    segs = (C.c_int32 * 2)(96, 40)
    h = C.c_void_p()
    L.check(lib.udt_pack_conv(wc.data_ptr(), bc.data_ptr(), 130, 136, 3, 3, segs, 2, 4, C.byref(h), st), "udt_pack_conv")
    rc = packing.pack_conv(wc, [96, 40])
    assert lib.udt_packed_dim(h, 3) == rc.shape[1] == 9 * (128 + 64)
    assert torch.equal(_packed_tensor(lib, h, "w", rc.shape, torch.bfloat16).view(torch.int16), rc.view(torch.int16))
    L.check(lib.udt_free_packed(h), "udt_free_packed")
    assert lib.udt_free_packed(None) == 0
