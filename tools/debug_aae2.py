"""block output_blocks.11 (two-source ResBlock 640 -> 320 + SpatialTransformer 320) alone, at the operating point of the G13 call:
HIP tape vs oracle autograd on the same inputs (the oracle's activations rounded to bf16), loss = the block's own t_attn local loss."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import udifftext_amd
from udifftext_amd import backward, pipeline, synth, ops
from oracle import nets, sampling as osamp, spec

dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
engine = pipeline.build_engine(dev)
unet = engine.model.diffusion_model
g13 = np.load("tests/golden/aae_golden.npz")
batch_cpu = synth.synthetic_batch(1, 128, 128, 4, seed=13)
torch.manual_seed(1234)
batch, buc = pipeline.prepare_batch(batch_cpu, dev)
c, uc = engine.conditioner.get_unconditional_conditioning(batch, batch_uc=buc, force_uc_zero_embeddings=["label"])
x = torch.from_numpy(g13["g13_x"])
sampler = pipeline.init_sampling(10, 5.0, dev)
c_noise = sampler.get_c_noise(x.to(dev), engine, torch.from_numpy(g13["g13_sigma"]).to(dev))
rel = lambda a, b: ((a.double() - b.double()).pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt().clamp_min(1e-300)).item()
sd = {k: v.detach().float().cpu() for k, v in engine.state_dict().items() if k.startswith("model.") or k.startswith("loss_fn.")}
cfg = spec.EngineConfig()
cc = {k: v.float().cpu() for k, v in c.items()}
mask, seg = batch["mask"].float().cpu(), batch["seg_mask"].float().cpu()
taps = {}
nets.unet_forward(sd, torch.cat((x, cc["concat"]), 1), c_noise.cpu(), cc["t_crossattn"], cfg.unet, taps=taps)
bf = lambda t: t.bfloat16().float()
h_in, skip = bf(taps["output_blocks.10"]), bf(taps["input_blocks.0"])
P = "model.diffusion_model."
emb = nets.timestep_embedding(c_noise.cpu(), cfg.unet.model_channels)
emb = nets._lin(sd, P + "time_embed.2.", F.silu(nets._lin(sd, P + "time_embed.0.", emb)))
ctx = bf(cc["t_crossattn"])
gk = sd["loss_fn.g_kernel"]
blk = unet.output_blocks[11]
rb, st = blk[0], blk[1]

def oracle(which):
    with torch.enable_grad():
        a, b = h_in.clone().requires_grad_(True), skip.clone().requires_grad_(True)
        maps = []
        h1 = nets._resblock(sd, P + "output_blocks.11.0.", torch.cat([a, b], 1), emb)
        h1.retain_grad()
        h2 = nets._spatial_transformer(sd, P + "output_blocks.11.1.", h1, ctx, 5, "o11.", maps)
        ll = osamp.min_local_loss(maps, mask, seg, gk, 1)
        ll.sum().backward()
    return ll.detach(), h1.detach(), h1.grad, a.grad, b.grad, maps[0]["attn_map"].detach()
ll_o, h1_o, dh1_o, da_o, db_o, pm_o = oracle(0)
nhwc = lambda t: t.to(dev).permute(0, 2, 3, 1).contiguous().bfloat16()
nchw = lambda t: t.float().permute(0, 3, 1, 2).cpu()
emb_rows = unet.time_embedding_rows(c_noise.float())
t_kv = unet.project_context(ctx.to(dev))
rec = []
h1, b_res = backward.resblock_fwd(rb, nhwc(h_in), emb_rows, x2=nhwc(skip))
h2, b_st = backward.spatial_transformer_fwd(st, h1, t_kv[st.st_index], rec, "o11.")
it = rec[0]
it["d_probs"] = torch.zeros_like(it["attn_map"])
loss = torch.zeros((1,), device=dev)
ops.local_loss_bwd(it["attn_map"], mask.to(dev), seg.to(dev), gk[0, 0].reshape(9).contiguous().to(dev), it["d_probs"], loss, 5, 16, 1.0)
print("loss", float(loss), float(ll_o), " h1 fwd", rel(nchw(h1), h1_o), " probs", rel(it["attn_map"].cpu(), pm_o))
d_h1 = b_st(None)
print("d wrt ST input (h1)      HIP vs oracle", rel(nchw(d_h1), dh1_o))
d_a, d_b = b_res(d_h1)
print("d wrt block input h      HIP vs oracle", rel(nchw(d_a), da_o), " skip", rel(nchw(d_b), db_o))
# the ResBlock's reverse pass alone, fed the ORACLE's cotangent
d_a2, d_b2 = b_res(nhwc(dh1_o))
print("ResBlock reverse alone (oracle cotangent in): h", rel(nchw(d_a2), da_o), " skip", rel(nchw(d_b2), db_o))
# ST internals: which piece of the chain loses it?  oracle sub-gradients by finite pieces
tb = st.transformer_blocks[0]
with torch.enable_grad():
    h1r = bf(h1_o).clone().requires_grad_(True)
    p = P + "output_blocks.11.1."
    b_, cch, hh, ww = h1r.shape
    t = nets._gn(sd, p + "norm.", h1r, 1e-6).permute(0, 2, 3, 1).reshape(b_, hh * ww, cch)
    t1 = nets._lin(sd, p + "proj_in.", t); t1.retain_grad()
    q = p + "transformer_blocks.0."
    ln = lambda n, tt: F.layer_norm(tt, (cch,), sd[q + n + ".weight"], sd[q + n + ".bias"], 1e-5)
    t2 = nets._self_attention(sd, q + "attn1.", ln("norm1", t1), 5) + t1; t2.retain_grad()
    maps = []
    t3 = nets._text_cross_attention(sd, q + "t_attn.", ln("t_norm", t2), ctx, 5, "x.t_attn", maps) + t2
    ll = osamp.min_local_loss(maps, mask, seg, gk, 1)
    ll.sum().backward()
print("oracle cotangent rms: t2", float(t2.grad.pow(2).mean().sqrt()), " t1", float(t1.grad.pow(2).mean().sqrt()), " h1", float(h1r.grad.pow(2).mean().sqrt()))
# HIP pieces on the oracle's tensors
M = hh * ww
t1_b, t2_b = bf(t1.detach()).reshape(M, cch).to(dev).bfloat16(), bf(t2.detach()).reshape(M, cch).to(dev).bfloat16()
# (a) from dP to d_t2
n2 = ops.layer_norm(t2_b, tb.t_norm.weight, tb.t_norm.bias, tb.t_norm.eps)
qh = tb.t_attn.to_q(n2).reshape(1, M, cch)
kv = t_kv[st.st_index][0]
probs = torch.empty((5, M, kv.shape[1]), dtype=torch.float32, device=dev)
ops.xattention(qh, kv[..., :cch], kv[..., cch:], 5, 64, tb.t_attn.scale, probs=probs)
dp = torch.zeros_like(probs); lz = torch.zeros((1,), device=dev)
ops.local_loss_bwd(probs, mask.to(dev), seg.to(dev), gk[0, 0].reshape(9).contiguous().to(dev), dp, lz, 5, 16, 1.0)
dq = ops.xattention_bwd(kv[..., :cch], kv[..., cch:], probs, dp, None, 5, tb.t_attn.scale)
d_n2 = backward.linear_bwd(tb.t_attn.to_q, dq.reshape(M, cch))
d_t2 = ops.layer_norm_bwd(t2_b, d_n2, tb.t_norm.weight, tb.t_norm.eps)
print("(a) dP -> d_t2 (xattn bwd, to_q bwd, t_norm bwd)  HIP vs oracle", rel(d_t2.float().cpu().reshape(1, M, cch), t2.grad))
# (b) from the ORACLE's d_t2 to d_t1 through attn1
d_t2o = bf(t2.grad).reshape(M, cch).to(dev).bfloat16()
a1 = tb.attn1
n1 = ops.layer_norm(t1_b, tb.norm1.weight, tb.norm1.bias, tb.norm1.eps)
qkv = ops.linear(n1, a1.packed()[0]).reshape(1, M, 3 * cch)
o = ops.attention_rowv(qkv[..., :cch], qkv[..., cch:2 * cch], qkv[..., 2 * cch:], 5, 0.125)
d_o = backward.linear_bwd(a1.to_out[0], d_t2o).reshape(1, M, cch)
d_qkv = ops.attention_bwd(qkv, o, d_o, 5, 0.125)
wq = torch.cat([a1.to_q.weight, a1.to_k.weight, a1.to_v.weight], 0).float()
d_n1 = (d_qkv.reshape(M, 3 * cch).float() @ wq).bfloat16()
d_t1 = ops.layer_norm_bwd(t1_b, d_n1, tb.norm1.weight, tb.norm1.eps, add=d_t2o)
print("(b) oracle d_t2 -> d_t1 (attn1 reverse)            HIP vs oracle", rel(d_t1.float().cpu().reshape(1, M, cch), t1.grad))
# the attention backward itself against autograd on the same qkv
with torch.enable_grad():
    tq = qkv.float().clone().requires_grad_(True)
    qq, kk_, vv_ = (tq[..., i * cch:(i + 1) * cch].reshape(1, M, 5, 64).permute(0, 2, 1, 3) for i in range(3))
    ro = (torch.softmax(qq @ kk_.transpose(-1, -2) * 0.125, dim=-1) @ vv_).permute(0, 2, 1, 3).reshape(1, M, cch)
    (rg,) = torch.autograd.grad((ro * d_o.float()).sum(), [tq])
for i, nm in enumerate("qkv"):
    print(f"    attention backward d{nm} at the operating point vs autograd", rel(d_qkv[..., i * cch:(i + 1) * cch].float(), rg[..., i * cch:(i + 1) * cch]),
          " rms", float(rg[..., i * cch:(i + 1) * cch].pow(2).mean().sqrt()))
vh = qkv[..., 2 * cch:].float()
print("    v: rms", float(vh.pow(2).mean().sqrt()), " rms of (v - token mean)", float((vh - vh.mean(dim=1, keepdim=True)).pow(2).mean().sqrt()))
