"""Where does the HIP attend-and-excite gradient leave the oracle's?  (GPU box; the oracle runs on the host cores)
Per counted t_attn layer: the (token, pixel) the hard min / max select, oracle vs HIP, with the margins; per block boundary: the
cotangent d loss / d (block output), HIP vs oracle autograd.    python tools/debug_aae.py [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import udifftext_amd
from udifftext_amd import backward, pipeline, synth
from oracle import nets, sampling as osamp, spec, backward as ob

dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
engine = pipeline.build_engine(dev)
g13 = np.load("tests/golden/aae_golden.npz")
sys.path.insert(0, "tests"); from aae_fixture import aae_batch
batch_cpu = aae_batch()
torch.manual_seed(1234)
batch, buc = pipeline.prepare_batch(batch_cpu, dev)
c, uc = engine.conditioner.get_unconditional_conditioning(batch, batch_uc=buc, force_uc_zero_embeddings=["label"])
if os.environ.get("AAE_OWN_COND", "0") == "0":      # the golden's conditioning (default): see tests/test_backward_gpu.py
    c = {"concat": torch.from_numpy(g13["g13_c_concat"]).to(dev), "t_crossattn": torch.from_numpy(g13["g13_c_txt"]).to(dev)}
x = torch.from_numpy(g13["g13_x"]).to(dev)
sampler = pipeline.init_sampling(10, 5.0, dev)
c_noise = sampler.get_c_noise(x, engine, torch.from_numpy(g13["g13_sigma"]).to(dev))
dbg = {}
loss, grad = backward.unet_local_loss_grad(engine.model.diffusion_model, engine.loss_fn, x, c_noise.float(), c["concat"], c["t_crossattn"],
                                           batch["mask"], batch["seg_mask"], debug=dbg)
rel = lambda a, b: ((a.double() - b.double()).pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt().clamp_min(1e-300)).item()
print("HIP loss", float(loss), "ref", float(g13["g13_local_loss"]), "grad rel", rel(grad.cpu(), torch.from_numpy(g13["g13_grad"])))

# oracle with taps
sd = {k: v.detach().float().cpu() for k, v in engine.state_dict().items() if k.startswith("model.") or k.startswith("loss_fn.") or k.startswith("denoiser.")}
cfg = spec.EngineConfig()
cc = {k: v.float().cpu() for k, v in c.items()}
mask, seg = batch["mask"].float().cpu(), batch["seg_mask"].float().cpu()
with torch.enable_grad():
    xg = x.cpu().clone().requires_grad_(True)
    maps, taps = [], {}
    nets.unet_forward(sd, torch.cat((xg, cc["concat"]), 1), c_noise.cpu(), cc["t_crossattn"], cfg.unet, attn_maps=maps, taps=taps)
    for t in taps.values():
        t.retain_grad()
    for m in maps:
        m["attn_map"].retain_grad()
    ll = osamp.min_local_loss(maps, mask, seg, sd["loss_fn.g_kernel"], 16)
    ll.sum().backward()
print("oracle loss", float(ll), "oracle grad vs golden", rel(xg.grad, torch.from_numpy(g13["g13_grad"])), " HIP grad vs oracle", rel(grad.cpu(), xg.grad))

def select(am, heads, size):
    seg_l = seg.shape[1]
    a = am.reshape(-1, heads, size * size, am.shape[-1])[..., :seg_l].permute(0, 1, 3, 2).mean(dim=1)
    a = F.conv2d(a.reshape(-1, seg_l, size, size), sd["loss_fn.g_kernel"], padding=1, groups=seg_l).reshape(-1, seg_l, size * size)
    mm = F.interpolate(mask, (size, size)).tile((1, seg_l, 1, 1)).reshape(-1, seg_l, size * size)
    v, idx = (mm * a).max(dim=-1)
    p = v + (1 - seg)
    srt = p[0].sort()
    top2 = (mm * a)[0, srt.indices[0]].sort(descending=True)
    return int(srt.indices[0]), int(idx[0, srt.indices[0]]), float(srt.values[1] - srt.values[0]), float(top2.values[0] - top2.values[1]), float(top2.values[0])

hip_maps = {m["name"]: m for m in dbg["maps"]}
for m in maps:
    if m["size"] < 16:
        continue
    hm = hip_maps[m["name"]]
    so = select(m["attn_map"].detach(), m["heads"], m["size"])
    sh = select(hm["attn_map"].float().cpu(), hm["heads"], hm["size"])
    print(f"{m['name']:55s} oracle (l*, n*, token margin, pixel margin, max) {so}\n{'':55s} HIP    {sh}   probs rel {rel(hm['attn_map'].cpu(), m['attn_map'].detach()):.2e}"
          f"  dP rel {rel(hm['d_probs'].cpu(), m['attn_map'].grad):.2e}")
for k in sorted(dbg):
    if k.startswith("d_"):
        name = k[2:]
        if name in taps and taps[name].grad is not None:
            got = dbg[k].float().permute(0, 3, 1, 2).cpu()
            print(f"cotangent at {name:20s} HIP vs oracle {rel(got, taps[name].grad):.3e}   (rms {taps[name].grad.pow(2).mean().sqrt():.2e})")
