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
runs = []
for r in range(3):
    backward.DEBUG_SUMS = []
    tape, noised, sigma = tr.training_tape(engine, z, cond, idx, noise)
    torch.cuda.synchronize()
    runs.append(list(backward.DEBUG_SUMS) + [("eps", float(tape.eps.abs().sum())), ("noised", float(noised.abs().sum()))])
    if r == 0:
        ld, gr = tr.training_loss_and_grads(engine, z, cond, seg, segm, sigma_idx=idx, noise=noise)   # one full step in between
        torch.cuda.synchronize()
    del tape
for i, (a, b, c) in enumerate(zip(*runs)):
    flag = "" if a[1] == b[1] == c[1] else "   <-- differs"
    print(f"{a[0]:32s} {a[1]:.6e} {b[1]:.6e} {c[1]:.6e}{flag}")
