import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np, torch
import udifftext_amd
from udifftext_amd import pipeline, training as tr, ops, backward
from aae_fixture import train_batch
dev = torch.device("cuda", 0); torch.set_grad_enabled(False)
engine = pipeline.build_engine(dev)
g = np.load("tests/golden/train_golden.npz")
batch = train_batch()
z, idx, noise = (torch.from_numpy(g[k]).to(dev) for k in ("g14_z", "g14_sigma_idx", "g14_noise"))
cond = {"concat": torch.from_numpy(g["g14_c_concat"]).to(dev), "t_crossattn": torch.from_numpy(g["g14_c_txt"]).to(dev)}
seg, segm = batch["seg"].to(dev), batch["seg_mask"].to(dev)

def snapshot():
    snap = {}
    for n, p in engine.model.state_dict().items():
        snap["sd:" + n] = p.detach().double().sum().item()
    for n, m in engine.model.named_modules():
        for attr in ("_pk", "_pkln", "_emb_w", "_emb_b"):
            v = getattr(m, attr, None)
            if v is None: continue
            for i, t in enumerate(v if isinstance(v, (tuple, list)) else [v]):
                if isinstance(t, torch.Tensor):
                    snap[f"{attr}:{n}:{i}"] = t.detach().double().sum().item()
    return snap
# forward-only first (packs everything lazily), then snapshot, then a grad call, then compare
ld0, _ = tr.training_loss_and_grads(engine, z, cond, seg, segm, sigma_idx=idx, noise=noise, want_grads=False)
ld0b, _ = tr.training_loss_and_grads(engine, z, cond, seg, segm, sigma_idx=idx, noise=noise, want_grads=False)
print("forward only twice", float(ld0["loss/diff_loss"]), float(ld0b["loss/diff_loss"]))
s0 = snapshot()
tape, noised, sigma = tr.training_tape(engine, z, cond, idx, noise)
loss_diff, d_eps = ops.diff_loss_grad(tape.eps, noised, z.float().contiguous(), sigma)
d = tape.backward(d_eps, param_grads=None)
s1 = snapshot()
print("after dX-only reverse pass: changed", [k for k in s0 if s0[k] != s1[k]][:10])
ld1, _ = tr.training_loss_and_grads(engine, z, cond, seg, segm, sigma_idx=idx, noise=noise, want_grads=False)
print("forward after dX-only reverse", float(ld1["loss/diff_loss"]))
tape, noised, sigma = tr.training_tape(engine, z, cond, idx, noise)
loss_diff, d_eps = ops.diff_loss_grad(tape.eps, noised, z.float().contiguous(), sigma)
pg = {}
d = tape.backward(d_eps, param_grads=pg)
s2 = snapshot()
ch = [k for k in s0 if s0[k] != s2[k]]
print("after reverse pass WITH parameter gradients: changed", len(ch), ch[:12])
ld2, _ = tr.training_loss_and_grads(engine, z, cond, seg, segm, sigma_idx=idx, noise=noise, want_grads=False)
print("forward after", float(ld2["loss/diff_loss"]))
