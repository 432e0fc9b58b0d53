"""BASELINE config #5, second generation: the MX8 (e4m3 elements + E8M0 block scales) linears of the transformer blocks — the
lean GEMM's FP8 / EMIT instances (csrc/lean.h), the MX8-emitting epilogues of the producers (lean GEMM, fused text
cross-attention, flash attention, GroupNorm apply) — against plain PyTorch fp32 restatements on the SAME quantised operands
(tests/mx8_ref.py decodes the kernels' format), through the C ABI.  ``pytest -m gpu``.

Reference ops: every nn.Linear of sgm/modules/attention.py:44-70 (GEGLU / FeedForward), :193-199 (to_q / to_k / to_v / to_out),
:375-411 (proj_in / proj_out), with the LayerNorms of :310-339 folded in.
"""
import math

import pytest
import torch
import torch.nn.functional as F

import mx8_ref

pytestmark = pytest.mark.gpu

# stated tolerances
#   GEMM on given e4m3 operands vs fp32 torch on the same dequantised operands: fp32 accumulation order + ONE bf16 rounding
REL_GEMM = 5e-3
#   an emitted MX8 activation vs the bf16 result it twins: e4m3 rounding, 3 mantissa bits: relative rms 2^-4 / sqrt(12) * ~1.4
REL_Q8 = 4e-2


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
        ops, packing, GEGLU, lib_mod = O, P, L.GEMM_GEGLU, L
    return Env


def _check_q8(q8, ref_bf16, stats=True):
    """an emitted Mx8Act against the bf16 tensor it twins: decoded values, block scales tight (amax / scale in (224, 448]),
    partial row statistics"""
    dec = mx8_ref.decode(q8.data, q8.scale)
    ref = ref_bf16.float()
    assert _rel(dec, ref) < REL_Q8
    M, K = ref.shape
    amax = ref.reshape(M, K // 32, 32).abs().amax(dim=2)
    qmax = q8.data.view(torch.float8_e4m3fn).float().reshape(M, K // 32, 32).abs().amax(dim=2)
    live = amax > 1e-20
    assert float(qmax[live].max()) <= 448.0
    assert float((qmax[live] > 200.0).float().mean()) > 0.99      # every block uses the top binades (bf16 vs fp32 amax: a few ulps)
    # element-wise: |dec - ref| <= half an e4m3 step at the block's scale (2^-4 of the value's binade, <= amax / 16) + the bf16 rounding
    # of the twin
    bound = (amax / 14.0)[:, :, None].expand(M, K // 32, 32).reshape(M, K) + ref.abs() * 2.0 ** -7 + 1e-30
    assert bool(((dec - ref).abs() <= bound).all())
    if stats and q8.stats is not None:
        s = q8.stats.sum(dim=0)
        assert torch.allclose(s[:, 0], ref.sum(dim=1), rtol=2e-2, atol=2e-2 * float(ref.abs().sum(dim=1).mean()))
        assert torch.allclose(s[:, 1], ref.pow(2).sum(dim=1), rtol=2e-2)


@pytest.mark.parametrize("M,N,K,res", [(2048, 1280, 1280, True), (8192, 640, 640, False), (520, 1280, 1280, True),
                                       (2048, 640, 2560, True)])
def test_bf16_linear_emits_mx8_twin_and_row_statistics(env, cuda, M, N, K, res):
    """proj_in in config #5: a bf16 GEMM whose epilogue also writes its result as an MX8 activation + partial row statistics"""
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn((M, K), generator=g).to(cuda).bfloat16()
    w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(cuda)
    b = torch.randn((N,), generator=g).to(cuda)
    r = (torch.randn((M, N), generator=g) * 3).to(cuda).bfloat16() if res else None
    # rows of very different magnitude: the block scales must follow them
    x = (x.float() * torch.logspace(-2, 2, M, device=cuda)[:, None]).bfloat16()
    wp = env.packing.pack_linear(w)
    plain = env.ops.linear(x, wp, b, residual=r)
    out = env.ops.linear(x, wp, b, residual=r, emit_q8=True, emit_rowstats=True)
    q8 = env.ops.mx8_of(out)
    assert q8 is not None and q8.stats is not None and q8.stats.shape[0] == N // 64
    assert torch.equal(out, plain)                               # the bf16 result is the non-emitting kernel's, bit for bit
    _check_q8(q8, out)


def _mx8_operands(env, dev, M, N, K, seed, geglu=False, ln=False):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn((M, K), generator=g).to(dev)
    x = x * torch.logspace(-1, 1, M, device=dev)[:, None] + (0.3 if ln else 0.0)
    w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(dev) * (1.0 + torch.arange(N, device=dev)[:, None] / N)
    b = torch.randn((N,), generator=g).to(dev)
    xq, xs = mx8_ref.encode(x)
    return x, mx8_ref.decode(xq, xs), xq, xs, w, b, g


@pytest.mark.parametrize("M,N,K,res,stats", [(2048, 1280, 5120, True, False), (8192, 640, 2560, True, False),
                                             (2048, 1280, 1280, True, True), (8192, 640, 640, False, True),
                                             (520, 1280, 1280, True, False), (200, 256, 128, False, False)])
def test_mx8_linear_vs_torch_on_the_same_quantised_operands(env, cuda, M, N, K, res, stats):
    """ff.net[2] / to_out / proj_out in config #5 (plain epilogue; proj_out also emits the next GroupNorm's column statistics)"""
    x, xd, xq, xs, w, b, g = _mx8_operands(env, cuda, M, N, K, seed=2)
    wq, cs = env.packing.pack_linear_fp8(w)
    wd = wq.view(torch.float8_e4m3fn).float() * cs[:, None]
    r = torch.randn((M, N), generator=g).to(cuda).bfloat16() if res else None
    ref = xd @ wd.t() + b + (r.float() if res else 0.0)
    act = env.ops.Mx8Act(xq, xs)
    rpb = M // 8 if (stats and M % 8 == 0) else 0
    out = env.ops.linear_mx8(act, wq, cs, b, residual=r, rows_per_batch=rpb, colstats=stats)
    assert _rel(out, ref) < REL_GEMM
    if stats:
        st = env.ops.gn_stats_of(out)
        assert st is not None
        tot = st.data.reshape(8, st.slots_per_sample, N, 2).sum(dim=1)
        o3 = out.float().reshape(8, M // 8, N)
        assert torch.allclose(tot[..., 0], o3.sum(dim=1), rtol=1e-3, atol=1e-2 * float(o3.abs().sum(dim=1).mean()))
    # the same launch with the MX8 twin of its result (ff.net[2] -> proj_out)
    if N % 128 == 0 and not stats:
        out2 = env.ops.linear_mx8(act, wq, cs, b, residual=r, emit_q8=True)
        assert torch.equal(out2, out)
        _check_q8(env.ops.mx8_of(out2), out2, stats=False)
    # against the UN-quantised product: the price of e4m3 operands (reported, loosely bounded)
    full = x @ w.t() + b + (r.float() if res else 0.0)
    print(f"mx8 linear {M}x{N}x{K}: vs same-operand fp32 {_rel(out, ref):.2e}, vs unquantised fp32 {_rel(out, full):.2e}")
    assert _rel(out, full) < 6e-2


@pytest.mark.parametrize("M,N,K", [(2048, 3840, 1280), (8192, 1920, 640), (520, 3840, 1280)])
def test_mx8_layernorm_folded_linear(env, cuda, M, N, K):
    """attn1's q|k|v in config #5: LayerNorm folded into an MX8 GEMM, the row statistics from the producer's partial sums"""
    x, xd, xq, xs, w, _, g = _mx8_operands(env, cuda, M, N, K, seed=3, ln=True)
    gamma = (1.0 + 0.2 * torch.randn((K,), generator=g)).to(cuda)
    beta = (0.1 * torch.randn((K,), generator=g)).to(cuda)
    wq, cs, c, s = env.packing.pack_ln_linear_mx8(w, None, gamma, beta)
    # partial statistics as a producer would emit them: P parts of K / P columns each, of the UN-quantised rows
    P = K // 64
    parts = torch.stack([x.reshape(M, P, 64).sum(dim=2).t(), x.reshape(M, P, 64).pow(2).sum(dim=2).t()], dim=2).contiguous()
    act = env.ops.Mx8Act(xq, xs, parts)
    out = env.ops.linear_mx8(act, wq, cs, ln_c=c, ln_s=s, eps=1e-5)
    mean = x.mean(dim=1, keepdim=True)
    rstd = (x.var(dim=1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    wd = wq.view(torch.float8_e4m3fn).float() * cs[:, None]
    ref = rstd * (xd @ wd.t() - mean * s[None, :]) + c[None, :]
    assert _rel(out, ref) < REL_GEMM
    full = F.layer_norm(x, (K,), gamma, beta, 1e-5) @ w.t()
    print(f"mx8 ln-linear {M}x{N}x{K}: vs same-operand fp32 {_rel(out, ref):.2e}, vs unquantised fp32 {_rel(out, full):.2e}")
    assert _rel(out, full) < 6e-2


@pytest.mark.parametrize("M,N,K", [(2048, 10240, 1280), (8192, 5120, 640), (520, 10240, 1280)])
def test_mx8_layernorm_folded_geglu_emits_mx8_hidden(env, cuda, M, N, K):
    """ff.net[0] in config #5: LayerNorm + GEGLU on MX8 operands, the hidden activation written as MX8 only"""
    x, xd, xq, xs, w, b, g = _mx8_operands(env, cuda, M, N, K, seed=4, ln=True)
    gamma = (1.0 + 0.2 * torch.randn((K,), generator=g)).to(cuda)
    beta = (0.1 * torch.randn((K,), generator=g)).to(cuda)
    wq, cs, c, s = env.packing.pack_ln_linear_mx8(w, b, gamma, beta, geglu=True)
    P = K // 64
    parts = torch.stack([x.reshape(M, P, 64).sum(dim=2).t(), x.reshape(M, P, 64).pow(2).sum(dim=2).t()], dim=2).contiguous()
    act = env.ops.Mx8Act(xq, xs, parts)
    both = env.ops.linear_mx8(act, wq, cs, ln_c=c, ln_s=s, flags=env.GEGLU, emit_q8=True)
    only = env.ops.linear_mx8(act, wq, cs, ln_c=c, ln_s=s, flags=env.GEGLU, emit_q8=True, want_bf16=False)
    mean = x.mean(dim=1, keepdim=True)
    rstd = (x.var(dim=1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    wd = wq.view(torch.float8_e4m3fn).float() * cs[:, None]
    y = rstd * (xd @ wd.t() - mean * s[None, :]) + c[None, :]
    inv = torch.empty(N, dtype=torch.long)
    inv[env.packing.geglu_permutation(N // 2)] = torch.arange(N)
    y = y[:, inv.to(cuda)]                                          # back to [x | gate] order
    ref = y[:, :N // 2] * F.gelu(y[:, N // 2:])
    assert _rel(both, ref) < 8e-3
    _check_q8(env.ops.mx8_of(both), both, stats=False)
    assert isinstance(only, env.ops.Mx8Act)
    assert torch.equal(only.data, env.ops.mx8_of(both).data) and torch.equal(only.scale, env.ops.mx8_of(both).scale)


def test_mx8_problems_outside_the_lean_family_are_refused(env, cuda):
    """UDT_GEMM_MX8 has no fallback kernel: a problem the lean plan declines is an error, not a silent bf16 read of e4m3 bytes"""
    x, xd, xq, xs, w, b, g = _mx8_operands(env, cuda, 256, 64, 128, seed=5)          # N <= 64: first-generation kernel territory
    wq, cs = env.packing.pack_linear_fp8(w)
    with pytest.raises(ValueError):
        env.ops.linear_mx8(env.ops.Mx8Act(xq, xs), wq, cs, b)


@pytest.mark.parametrize("B,N,heads,zero", [(8, 1024, 10, 4), (8, 256, 20, 4), (2, 64, 20, 1), (8, 1024, 10, 0)])
def test_fused_text_attention_emits_mx8_twin_and_row_statistics(env, cuda, B, N, heads, zero):
    """x + t_attn(LN(x)) in config #5: the fused launch also writes its result as MX8 + the row statistics of norm3 — for the live
    samples and for the zero-context half, with and without the channel split (few token tiles)"""
    from udifftext_amd import lib as L
    O = env.ops
    C, Lc = heads * 64, 9
    g = torch.Generator(device="cpu").manual_seed(6)
    x = (torch.randn((B, N, C), generator=g) * torch.logspace(-1, 1, C)[None, None, :]).to(cuda).bfloat16()
    kv = torch.randn((B, Lc, 2 * C), generator=g).to(cuda).bfloat16()
    wq = env.packing.pack_linear((torch.randn((C, C), generator=g) / math.sqrt(C)).to(cuda))
    wo = env.packing.pack_linear((torch.randn((C, C), generator=g) / math.sqrt(C)).to(cuda))
    gamma, beta = (1 + 0.1 * torch.randn((C,), generator=g)).to(cuda), (0.1 * torch.randn((C,), generator=g)).to(cuda)
    bias = torch.randn((C,), generator=g).to(cuda)
    tables = O.tattn_prepare(kv, wq, wo, gamma, beta, heads, 0.125)
    plain = O.tattn_fused(x, tables, bias, heads, zero, 1e-5)
    out = O.tattn_fused(x, tables, bias, heads, zero, 1e-5, emit_q8=True)
    q8 = O.mx8_of(out)
    assert q8 is not None and q8.stats is not None
    assert q8.stats.shape[0] == L.load().udt_tattn_rowstat_parts(B, N, C)
    assert torch.equal(out, plain)
    _check_q8(q8, out.reshape(B * N, C))


@pytest.mark.parametrize("B,N,heads", [(8, 1024, 10), (8, 256, 20), (3, 200, 10)])
def test_flash_attention_emits_mx8_twin(env, cuda, B, N, heads):
    """attn1's flash kernel in config #5: O also as an MX8 activation for the e4m3 to_out"""
    O = env.ops
    C = heads * 64
    g = torch.Generator(device="cpu").manual_seed(7)
    qkv = torch.randn((B, N, 3 * C), generator=g).to(cuda).bfloat16()
    plain = O.attention_rowv(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, 0.125)
    out = O.attention_rowv(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, 0.125, emit_q8=True)
    q8 = O.mx8_of(out)
    assert q8 is not None and torch.equal(out, plain)
    _check_q8(q8, out.reshape(B * N, C), stats=False)


def test_mx8_launches_are_bit_reproducible(env, cuda):
    """the MX8 GEMM with a ticket split-K (2048 x 1280 x 5120: three K slices summed in slice order), the emitting epilogue and the
    fused text attention's MX8 twin give the same bits on every run — no atomics, no arrival-order sums"""
    x, xd, xq, xs, w, b, g = _mx8_operands(env, cuda, 2048, 1280, 5120, seed=8)
    wq, cs = env.packing.pack_linear_fp8(w)
    r = torch.randn((2048, 1280), generator=g).to(cuda).bfloat16()
    act = env.ops.Mx8Act(xq, xs)
    outs = [env.ops.linear_mx8(act, wq, cs, b, residual=r, emit_q8=True, emit_rowstats=True) for _ in range(3)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
        assert torch.equal(env.ops.mx8_of(o).data, env.ops.mx8_of(outs[0]).data)
        assert torch.equal(env.ops.mx8_of(o).scale, env.ops.mx8_of(outs[0]).scale)
        assert torch.equal(env.ops.mx8_of(o).stats, env.ops.mx8_of(outs[0]).stats)


# ------------------------------------------------------------------------------------------- e4m3 self-attention (config #5)
#   the e4m3 attention vs fp32 torch on the same dequantised q, k, v: P is rounded to e4m3 inside the kernel — a relative error of up
#   to 2^-4 per probability, rms 2.6e-2.  With ZERO-MEAN random v the output is itself a cancelling sum (|O| ~ |v| / sqrt(keys that
#   carry weight)), so nothing averages down and the output's relative error IS that rms; with v of one sign it averages down
REL_ATTN8 = 3.5e-2
REL_ATTN8_ONE_SIGN = 1.2e-2
V_MUL = 32.0


def _check_qkv8(q8, ref_bf16, C, v_mul):
    """the q|k|v projection's MX8 form: q and k thirds block-scaled, the v third e4m3(v * v_mul) with saturation"""
    M = ref_bf16.shape[0]
    qk = type(q8)(q8.data[:, :2 * C].contiguous(), q8.scale[: 2 * C // 128].contiguous())
    _check_q8(qk, ref_bf16[:, :2 * C], stats=False)
    v = ref_bf16[:, 2 * C:3 * C].float()
    got = q8.data[:, 2 * C:3 * C].contiguous().view(torch.float8_e4m3fn).float() / v_mul
    want = (v * v_mul).clamp(-448.0, 448.0)
    assert bool(((got * v_mul - want).abs() <= want.abs() * 2.0 ** -4 + 2.0 ** -10 + v.abs() * v_mul * 2.0 ** -7).all())
    assert _rel(got, v.clamp(-448.0 / v_mul, 448.0 / v_mul)) < REL_Q8


@pytest.mark.parametrize("M,C", [(8192, 320), (520, 320)])
def test_qkv_projection_of_the_row_resident_kernel_emits_mx8_with_a_fixed_scale_v(env, cuda, M, C):
    """attn1's q|k|v at the 320-channel level in config #5: the bf16 LayerNorm-folded K = 320 kernel writes ONLY the MX8 form"""
    g = torch.Generator(device="cpu").manual_seed(11)
    x = (torch.randn((M, C), generator=g).to(cuda) * torch.logspace(-1, 1, M, device=cuda)[:, None] + 0.3).bfloat16()
    w = (torch.randn((3 * C, C), generator=g) / math.sqrt(C)).to(cuda)
    w[2 * C:] *= 0.25
    gamma = (1.0 + 0.2 * torch.randn((C,), generator=g)).to(cuda)
    beta = (0.1 * torch.randn((C,), generator=g)).to(cuda)
    wp, c, s = env.packing.pack_ln_linear(w, None, gamma, beta)
    plain = env.ops.ln_linear(x, wp, c, s)
    q8 = env.ops.ln_linear(x, wp, c, s, emit_q8=True, want_bf16=False, q8_fixed=(2 * C, V_MUL))
    assert isinstance(q8, env.ops.Mx8Act)
    _check_qkv8(q8, plain, C, V_MUL)
    both = env.ops.ln_linear(x, wp, c, s, emit_q8=True, q8_fixed=(2 * C, V_MUL))
    # (below a chip-filling row count the plain launch runs on the tiled kernel, the emitting one always on the row-resident one)
    assert _rel(both, plain) < 4e-3
    q8b = env.ops.mx8_of(both)
    assert torch.equal(q8b.data, q8.data) and torch.equal(q8b.scale[: 2 * C // 128], q8.scale[: 2 * C // 128])


@pytest.mark.parametrize("M", [520, 264, 300, 8192 + 40])
def test_row_resident_emitting_kernel_with_ragged_last_tile_is_repeatable(env, cuda, M):
    """round 6 (ADVICE): the emitting epilogue's stores sit under the row predicate, so a wave whose 64 rows are partly / wholly past
    M issues fewer of them; its `vmcnt` count at the chunk barrier must not assume the full 16 (it would let the other waves read a
    weight chunk whose LDS-DMA has not landed).  The rows of the ragged tile against the same rows computed inside a FULL tile (the
    kernel is row-independent), many times over, while a second stream keeps the memory system busy."""
    C = 320
    g = torch.Generator(device="cpu").manual_seed(21)
    Mfull = ((M + 255) // 256) * 256
    x = (torch.randn((Mfull, C), generator=g).to(cuda) + 0.3).bfloat16()
    w = (torch.randn((3 * C, C), generator=g) / math.sqrt(C)).to(cuda)
    gamma = (1.0 + 0.2 * torch.randn((C,), generator=g)).to(cuda)
    beta = (0.1 * torch.randn((C,), generator=g)).to(cuda)
    wp, c, s = env.packing.pack_ln_linear(w, None, gamma, beta)
    ref = env.ops.ln_linear(x, wp, c, s, emit_q8=True, want_bf16=False, q8_fixed=(2 * C, V_MUL))
    xr = x[:M].contiguous()
    noise = torch.empty((64 << 20,), device=cuda, dtype=torch.uint8)
    side = torch.cuda.Stream()
    for it in range(12):
        with torch.cuda.stream(side):
            noise.add_(1)                                  # (competing traffic: delays the LDS-DMA of the kernel under test)
        q8 = env.ops.ln_linear(xr, wp, c, s, emit_q8=True, want_bf16=False, q8_fixed=(2 * C, V_MUL))
        assert torch.equal(q8.data, ref.data[:M]), it
        assert torch.equal(q8.scale[: 2 * C // 128, :M], ref.scale[: 2 * C // 128, :M]), it     # (whole dwords of q and k)
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,C", [(2048, 640), (520, 1280), (2048, 1280)])
def test_mx8_layernorm_folded_qkv_projection_emits_mx8_with_a_fixed_scale_v(env, cuda, M, C):
    """the same at the 640 / 1280-channel levels: MX8 in (the block's producer), MX8 out (for the e4m3 attention)"""
    x, xd, xq, xs, w, _, g = _mx8_operands(env, cuda, M, 3 * C, C, seed=12, ln=True)
    w[2 * C:] *= 0.25
    gamma = (1.0 + 0.2 * torch.randn((C,), generator=g)).to(cuda)
    beta = (0.1 * torch.randn((C,), generator=g)).to(cuda)
    wq, cs, c, s = env.packing.pack_ln_linear_mx8(w, None, gamma, beta)
    P = C // 64
    parts = torch.stack([x.reshape(M, P, 64).sum(dim=2).t(), x.reshape(M, P, 64).pow(2).sum(dim=2).t()], dim=2).contiguous()
    act = env.ops.Mx8Act(xq, xs, parts)
    plain = env.ops.linear_mx8(act, wq, cs, ln_c=c, ln_s=s, eps=1e-5)
    q8 = env.ops.linear_mx8(act, wq, cs, ln_c=c, ln_s=s, eps=1e-5, emit_q8=True, want_bf16=False, q8_fixed=(2 * C, V_MUL))
    assert isinstance(q8, env.ops.Mx8Act)
    _check_qkv8(q8, plain, C, V_MUL)


def _qkv8(dev, B, N, heads, seed, sharp=1.0, one_sign=False):
    C = heads * 64
    g = torch.Generator(device="cpu").manual_seed(seed)
    qkv = torch.randn((B * N, 3 * C), generator=g).to(dev)
    qkv[:, :C] *= sharp * (0.5 + torch.rand((B * N, 1), generator=g).to(dev) * 2.0)       # rows of different magnitude
    qkv[:, 2 * C:] *= 0.3
    if one_sign:
        qkv[:, 2 * C:] += 1.0
    Cp = -(-2 * C // 128) * 128
    qk = F.pad(qkv[:, :2 * C], (0, Cp - 2 * C))
    qk8, qks = mx8_ref.encode(qk)
    v8 = (qkv[:, 2 * C:] * V_MUL).clamp(-448, 448).to(torch.float8_e4m3fn)
    data = torch.cat([qk8[:, :2 * C], v8.view(torch.uint8)], dim=1).contiguous()
    nt = (3 * C + 127) // 128
    scale = torch.zeros((nt, B * N), dtype=torch.int32, device=dev)
    scale[: qks.shape[0]] = qks
    qkd = mx8_ref.decode(qk8, qks)[:, :2 * C]
    return data, scale, qkd[:, :C], qkd[:, C:], v8.float() / V_MUL


@pytest.mark.parametrize("B,N,heads,sharp", [(2, 4096, 5, 1.0), (2, 1024, 10, 1.0), (4, 256, 20, 1.0), (3, 200, 10, 1.0),
                                             (2, 1024, 10, 6.0), (1, 64, 2, 1.0), (2, 4, 20, 1.0)])
def test_mx8_self_attention_vs_torch_on_the_same_quantised_operands(env, cuda, B, N, heads, sharp):
    """attn1 in config #5 with e4m3 Q K^T and P V (udt_attn_mx8_fwd); ragged key counts, one-tile and sub-tile sequences,
    peaked softmaxes (sharp)"""
    O = env.ops
    C = heads * 64
    data, scale, qd, kd, vd = _qkv8(cuda, B, N, heads, seed=13, sharp=sharp)
    out = O.attention_mx8(O.Mx8Act(data, scale), B, heads, 0.125, V_MUL, emit_q8=(heads % 2 == 0))

    def split(t):
        return t.reshape(B, N, heads, 64).permute(0, 2, 1, 3)
    p = torch.softmax(split(qd) @ split(kd).transpose(-1, -2) * 0.125, dim=-1)
    ref = (p @ split(vd)).permute(0, 2, 1, 3).reshape(B, N, C)
    err = _rel(out, ref)
    print(f"mx8 attention B={B} N={N} heads={heads} sharp={sharp}: vs same-operand fp32 {err:.2e}")
    assert err < REL_ATTN8
    q8 = O.mx8_of(out)
    if heads % 2 == 0:
        _check_q8(q8, out.reshape(B * N, C), stats=False)
    again = O.attention_mx8(O.Mx8Act(data, scale), B, heads, 0.125, V_MUL)
    assert torch.equal(again, out)
    # v of one sign: the rounding of P averages down over the keys
    data, scale, qd, kd, vd = _qkv8(cuda, B, N, heads, seed=15, sharp=sharp, one_sign=True)
    out = O.attention_mx8(O.Mx8Act(data, scale), B, heads, 0.125, V_MUL)
    p = torch.softmax(split(qd) @ split(kd).transpose(-1, -2) * 0.125, dim=-1)
    ref = (p @ split(vd)).permute(0, 2, 1, 3).reshape(B, N, C)
    print(f"    v of one sign: {_rel(out, ref):.2e}")
    assert _rel(out, ref) < REL_ATTN8_ONE_SIGN


def test_fixed_scale_columns_must_be_whole_blocks(env, cuda):
    """udt_gemm_q8_ok: a fixed-scale range that starts inside a 32-column block, a non-positive multiplier, or one on a launch that
    does not emit, has no plan (the MX8 GEMM refuses it instead of writing a half-scaled block)"""
    O, L = env.ops, env.lib_mod
    import ctypes as C
    M, K, N = 256, 128, 256
    q, sc = mx8_ref.encode(torch.randn((M, K), device=cuda))
    w = torch.randn((N, K), device=cuda) / math.sqrt(K)
    wq, cs = env.packing.pack_linear_fp8(w)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=cuda)
    q8 = O._mx8_alloc(M, N, cuda)

    def ok(col, mul):
        d = O.gemm_desc(a=q.data_ptr(), w=wq.data_ptr(), out=out.data_ptr(), M=M, N=N, K=K, lda=K, ldo=N, flags=L.GEMM_MX8,
                        colscale=cs.data_ptr(), a_scale=sc.data_ptr(), q8_out=q8.data.data_ptr(), q8_scale=q8.scale.data_ptr(), ld_q8=N,
                        q8_fixed_col=col, q8_fixed_mul=mul)
        return bool(L.load().udt_gemm_q8_ok(C.byref(d)))
    assert ok(0, 0.0) and ok(128, 32.0) and ok(160, 0.5)
    assert not ok(136, 32.0) and not ok(128, 0.0) and not ok(128, -1.0)


def test_mx8_self_attention_refuses_what_it_cannot_run(env, cuda):
    """status codes of the C entry, no launch: a row pitch below 3 C, an unaligned pitch, a missing scale array, v_inv <= 0"""
    data, scale, *_ = _qkv8(cuda, 1, 64, 2, seed=14)
    out = torch.empty((1, 64, 128), dtype=torch.bfloat16, device=cuda)
    lib = env.lib_mod.load()

    def call(ld8=384, sc=scale.data_ptr(), v_inv=1.0 / V_MUL, n=64):
        return lib.udt_attn_mx8_fwd(data.data_ptr(), sc, out.data_ptr(), 1, 2, n, ld8, 128, 0.125, v_inv, None, None, 0, None)
    assert call() == 0
    assert call(ld8=368) == -1 and call(ld8=392) == -1 and call(n=0) == -1
    assert call(sc=None) == -2 and call(v_inv=0.0) == -2


def test_mx8_self_attention_properties_at_full_size(env, cuda):
    """size-independent properties at the benchmarked size (8 samples x 4096 tokens x 5 heads: no torch reference of 2.7 GB of scores):
    (1) the denominator is the sum of exactly the numerators the P V product uses: with v CONSTANT along the keys (a different
        e4m3-exact constant per head dim) every output row equals those constants to one bf16 rounding, whatever q and k are;
    (2) attention does not care about the ORDER of the keys: k and v rows permuted together give the same output up to the rounding
        steps that depend on tile order (deferred maximum -> the exponent of a probability's byte, fp32 summation order);
    (3) samples are independent: sample 3 alone gives the same bits as inside the batch of 8."""
    O = env.ops
    B, N, heads = 8, 4096, 5
    C = heads * 64
    data, scale, *_ = _qkv8(cuda, B, N, heads, seed=21, sharp=2.0)
    consts = ((torch.arange(C, device=cuda) % 8 + 1).float() / 4.0)                    # 0.25 .. 2.0: x 32 = 8 .. 64, exact in e4m3
    vconst = (consts * V_MUL).to(torch.float8_e4m3fn).view(torch.uint8)
    d1 = data.clone()
    d1[:, 2 * C:] = vconst[None, :]
    out = O.attention_mx8(O.Mx8Act(d1, scale), B, heads, 0.125, V_MUL)
    err = (out.float() - consts[None, None, :]).abs() / consts[None, None, :]
    assert float(err.max()) <= 2.0 ** -8, float(err.max())
    # (2) (v of one sign: with zero-mean v the output is a cancelling sum that carries the probabilities' rounding noise in full, and a
    #      different key order is a fresh realisation of it — REL_ATTN8 times sqrt 2)
    data, scale, *_ = _qkv8(cuda, B, N, heads, seed=22, sharp=2.0, one_sign=True)
    out0 = O.attention_mx8(O.Mx8Act(data, scale), B, heads, 0.125, V_MUL)
    perm = torch.randperm(N, device=cuda)
    rows = (torch.arange(B, device=cuda)[:, None] * N + perm[None, :]).reshape(-1)
    d2 = data.clone()
    d2[:, C:] = data[rows][:, C:]                                                     # k and v of every sample in permuted key order
    s2 = scale.clone()
    kb = C // 128                                                                      # dwords that hold k's scale bytes: blocks [C/32, 2C/32)
    # a scale dword holds 4 blocks and q's / k's blocks may share one (C = 320: block 10 starts mid-dword): move k's BYTES
    sb = scale.view(torch.uint8).reshape(scale.shape[0], B * N, 4).permute(1, 0, 2).reshape(B * N, -1).clone()      # [row, block]
    sb2 = sb.clone()
    sb2[:, C // 32:2 * C // 32] = sb[rows][:, C // 32:2 * C // 32]
    s2 = sb2.reshape(B * N, scale.shape[0], 4).permute(1, 0, 2).contiguous().view(torch.int32).reshape(scale.shape)
    out2 = O.attention_mx8(O.Mx8Act(d2, s2), B, heads, 0.125, V_MUL)
    assert _rel(out2, out0) < 1.5 * REL_ATTN8_ONE_SIGN, _rel(out2, out0)
    # (3)
    one = O.attention_mx8(O.Mx8Act(data[3 * N:4 * N].contiguous(), scale[:, 3 * N:4 * N].contiguous()), 1, heads, 0.125, V_MUL)
    assert torch.equal(one[0], out0[3])
