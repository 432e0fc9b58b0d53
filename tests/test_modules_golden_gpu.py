"""The 16 G10 module goldens (tests/golden/module_golden.npz: outputs of the REAL reference's ResBlock / Upsample / Downsample /
MemoryEfficientCrossAttention / CrossAttention / FeedForward / BasicTransformerBlock / SpatialTransformer and of the VAE's
ResnetBlock / MemoryEfficientAttnBlock / Downsample / Upsample at small shapes, make_golden.py) against the HIP-backed modules of the
same names (udifftext_amd/sgm/modules) with the same name-keyed synthetic weights — module by module, not only through whole-network
outputs.  ``pytest -m gpu``.  Stated tolerance: bf16 storage / MFMA with fp32 accumulation against the fp32 reference: rel RMS
<= 2e-2 per module (attention probabilities 3e-2 of their RMS).

Reference: sgm/modules/diffusionmodules/openaimodel.py:89-146,183-268; sgm/modules/attention.py:44-70,111-174,177-262,265-341,344-415;
sgm/modules/diffusionmodules/model.py:64-88,111-148,214-262.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 2e-2


def _rel(got, ref):
    got, ref = torch.as_tensor(got).double().cpu(), torch.as_tensor(ref).double().cpu()
    return ((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-30)).item()


@pytest.fixture(scope="module")
def mg():
    return np.load(os.path.join(GOLD, "module_golden.npz"))


@pytest.fixture(scope="module")
def env(cuda):
    import udifftext_amd  # noqa: F401  (puts the sgm package on the path)
    from udifftext_amd import lib, ops, synth
    assert lib.load().udt_device_arch_ok() == 1
    torch.set_grad_enabled(False)

    class Env:
        pass
    Env.ops, Env.synth, Env.dev = ops, synth, cuda

    def build(name, m):
        m = m.eval()
        synth.fill_module_(m, prefix=f"g10.{name}.")
        return m.to(cuda)
    Env.build = staticmethod(build)
    Env.nhwc = staticmethod(lambda a: torch.from_numpy(a).to(cuda).permute(0, 2, 3, 1).contiguous().bfloat16())
    Env.nchw = staticmethod(lambda t: t.float().permute(0, 3, 1, 2).cpu())
    return Env


def test_g10_resblock_up_down(env, mg):
    from sgm.modules import hipnn as H
    from sgm.modules.diffusionmodules.openaimodel import Downsample, ResBlock, Upsample
    x = env.nhwc(mg["in_res_x"])
    emb = torch.from_numpy(mg["in_res_emb"]).to(env.dev)
    for name, cout in (("res_64_128", 128), ("res_64_64", 64)):
        rb = env.build(name, ResBlock(64, 256, 0.0, out_channels=cout))
        # the UNet evaluates all emb_layers as one GEMM on SiLU(emb); alone, the block's own Linear gives its rows
        rb.emb_offset = 0
        rows = rb.emb_layers[1](F.silu(emb).bfloat16(), flags=H.GEMM_OUT_F32)
        got = env.nchw(rb(x, rows))
        assert _rel(got, mg[name]) < TOL, (name, _rel(got, mg[name]))
    got = env.nchw(env.build("up_64", Upsample(64, True))(x))
    assert _rel(got, mg["up_64"]) < TOL
    got = env.nchw(env.build("down_64", Downsample(64, True))(x))
    assert _rel(got, mg["down_64"]) < TOL


def test_g10_attention_modules(env, mg):
    from sgm.modules.attention import BasicTransformerBlock, CrossAttention, FeedForward, MemoryEfficientCrossAttention, SpatialTransformer
    dev = env.dev
    t = torch.from_numpy(mg["in_tokens"]).to(dev).bfloat16()
    ctx = torch.from_numpy(mg["in_ctx"]).to(dev)
    # the context projections' K (96) is packed to a multiple of 64: pad the context's channels with zeros likewise
    ctxp = F.pad(ctx, (0, 128 - ctx.shape[-1])).bfloat16().contiguous()
    got = env.build("selfattn_128", MemoryEfficientCrossAttention(128, heads=2, dim_head=64))(t)
    assert _rel(got, mg["selfattn_128"]) < TOL
    ca = env.build("xattn_128", CrossAttention(128, context_dim=96, heads=2, dim_head=64))
    ca.attn_map_cache = {"size": None, "attn_map": None}
    got = ca(t, context=ctxp, emit_map=True)
    assert _rel(got, mg["xattn_128"]) < TOL
    assert ca.attn_map_cache["size"] == 8 and tuple(ca.attn_map_cache["attn_map"].shape) == mg["xattn_128_map"].shape
    assert _rel(ca.attn_map_cache["attn_map"], mg["xattn_128_map"]) < 3e-2
    got = ca(t, context=ctxp[:, :1].contiguous())                          # a single context token: the sigmoid branch
    assert _rel(got, mg["xattn_128_single"]) < TOL
    ff = env.build("ff_128", FeedForward(128, glu=True))
    got = ff(t.reshape(-1, 128)).reshape(2, 64, 128)
    assert _rel(got, mg["ff_128"]) < TOL
    blk = env.build("block_128", BasicTransformerBlock(128, 2, 64, t_context_dim=96))
    got = blk(t, t_context=ctxp)
    assert _rel(got, mg["block_128"]) < TOL
    st = env.build("st_128", SpatialTransformer(128, 2, 64, depth=1, t_context_dim=96, use_linear=True))
    got = env.nchw(st(env.nhwc(mg["in_st_x"]), t_context=ctxp))
    assert _rel(got, mg["st_128"]) < TOL


def test_g10_vae_modules(env, mg):
    from sgm.modules.diffusionmodules.model import Downsample, MemoryEfficientAttnBlock, ResnetBlock, Upsample
    x = env.nhwc(mg["in_vae_x"])
    for name, cout in (("vres_64_128", 128), ("vres_64_64", 64)):
        m = env.build(name, ResnetBlock(in_channels=64, out_channels=cout, dropout=0.0, temb_channels=0))
        got = env.nchw(m(x, None))
        assert _rel(got, mg[name]) < TOL, (name, _rel(got, mg[name]))
    got = env.nchw(env.build("vattn_64", MemoryEfficientAttnBlock(64))(x))
    assert _rel(got, mg["vattn_64"]) < TOL
    got = env.nchw(env.build("vdown_64", Downsample(64, True))(x))
    assert _rel(got, mg["vdown_64"]) < TOL
    got = env.nchw(env.build("vup_64", Upsample(64, True))(x))
    assert _rel(got, mg["vup_64"]) < TOL
