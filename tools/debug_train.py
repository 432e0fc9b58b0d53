import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np, torch
import udifftext_amd
from udifftext_amd import pipeline, training as tr, ops
from aae_fixture import train_batch
dev = torch.device("cuda", 0); torch.set_grad_enabled(False)
engine = pipeline.build_engine(dev)
g = np.load("tests/golden/train_golden.npz")
batch = train_batch()
z, idx, noise = (torch.from_numpy(g[k]).to(dev) for k in ("g14_z", "g14_sigma_idx", "g14_noise"))
cond = {"concat": torch.from_numpy(g["g14_c_concat"]).to(dev), "t_crossattn": torch.from_numpy(g["g14_c_txt"]).to(dev)}
seg, segm = batch["seg"].to(dev), batch["seg_mask"].to(dev)
rel = lambda a, b: ((a.double() - b.double()).pow(2).sum().sqrt() / b.double().pow(2).sum().sqrt().clamp_min(1e-300)).item()
runs = []
for lam in (0.01, 0.01, 0.0, 0.01):
    engine.loss_fn.lambda_local_loss = lam
    ld, gr = tr.training_loss_and_grads(engine, z, cond, seg, segm, sigma_idx=idx, noise=noise)
    runs.append(gr)
    print("lam", lam, {k: float(v) for k, v in ld.items()})
names = sorted(runs[0])
for i in (1, 2, 3):
    worst = sorted(((rel(runs[i][n], runs[0][n]), n) for n in names), reverse=True)[:4]
    tot = (sum(float((runs[i][n].double() - runs[0][n].double()).pow(2).sum()) for n in names) / sum(float(runs[0][n].double().pow(2).sum()) for n in names)) ** 0.5
    print("run", i, "vs run 0: total", tot, "worst", worst)
