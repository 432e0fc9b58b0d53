"""Where does a sample's UNet result start to depend on the batch it is in?  One call on the sampling loop's path for sample 2 alone and
inside a batch of 4; forward hooks on every top-level block print the relative RMS distance of that sample's activations (bf16 path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import udifftext_amd
from udifftext_amd import pipeline, synth, ops
from test_engine_gpu import _sampler_call
dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
eng = pipeline.build_engine(dev)
unet = eng.model.diffusion_model
torch.manual_seed(35)
B = 4
ctx = eng.conditioner.embedders[0](synth.synthetic_batch(B, 512, 512, 9, seed=16)["label"])
x = torch.randn((B, 9, 64, 64), device=dev)
rec = {}
def hook(name):
    def f(mod, inp, out):
        o = out[0] if isinstance(out, (tuple, list)) else out
        rec.setdefault(name, []).append(o.detach().float().clone())
    return f
names = []
for grp in ("input_blocks", "output_blocks"):
    for i, blk in enumerate(getattr(unet, grp)):
        for j, sub in enumerate(blk):
            n = f"{grp}.{i}.{j}:{type(sub).__name__}"
            names.append(n); sub.register_forward_hook(hook(n))
for j, sub in enumerate(unet.middle_block):
    n = f"middle.{j}:{type(sub).__name__}"; names.append(n); sub.register_forward_hook(hook(n))
st = unet.input_blocks[1][1]
for n, sub in (("st.norm", st.norm), ("st.proj_in", st.proj_in), ("st.block0", st.transformer_blocks[0]), ("st.block0.attn1", st.transformer_blocks[0].attn1),
               ("st.block0.attn1.to_out", st.transformer_blocks[0].attn1.to_out[0]), ("st.block0.ff", st.transformer_blocks[0].ff),
               ("st.block0.ff.net0", st.transformer_blocks[0].ff.net[0]), ("st.block0.ff.net2", st.transformer_blocks[0].ff.net[2]), ("st.proj_out", st.proj_out)):
    names.append(n); sub.register_forward_hook(hook(n))
_real_tattn = ops.tattn_fused
def _spy_tattn(xx, *a, **k):
    o = _real_tattn(xx, *a, **k)
    rec.setdefault("tattn_fused(first)", []).append(o.detach().float().clone()) if len(rec.get("tattn_fused(first)", [])) < 2 and xx.shape[-1] == 320 and not any(n == "done" for n in ()) else None
    return o
def call(idx):
    n = len(idx)
    xs = torch.cat([x[idx], x[idx]])
    tc = torch.cat([torch.zeros_like(ctx[idx]), ctx[idx]])
    return _sampler_call(unet, xs, torch.full((2 * n,), 441, device=dev), tc, n)
e4 = call([0, 1, 2, 3]); e1 = call([2])
def sel(a4, a1):
    # rows of sample 2 (uc, c) inside the batch of 4: tensors are [2B, ...] or [2B * N, C]
    if a4.shape[0] == 8: return a4[[2, 6]]
    n = a4.shape[0] // 8
    return torch.cat([a4[2 * n:3 * n], a4[6 * n:7 * n]])
def rel(a, b): return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30)).item()
for n in names:
    if n in rec and len(rec[n]) == 2:
        a4, a1 = rec[n]
        print(f"{n:55s} {rel(a1, sel(a4, a1)):.3e}   shape {tuple(a1.shape)}")
print("eps", rel(e1.float(), e4[[2, 6]].float()))
