"""A/B inside one process: sampler step as hipGraph replay, one launch stream vs uc/c halves on two streams"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import pipeline, synth
import sgm.modules.diffusionmodules.sampling as S
dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
B, size, NS = int(os.environ.get("B", 4)), int(os.environ.get("SIZE", 512)), 8
model = pipeline.build_engine(dev)
sampler = pipeline.init_sampling(50, 5.0, dev)
sig = sampler._host_sigmas()
b = synth.synthetic_batch(B, size, size, 9, seed=1)
b = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
batch, buc = pipeline.prepare_batch(b, dev)
c, uc = model.conditioner.get_unconditional_conditioning(batch, batch_uc=buc, force_uc_zero_embeddings=["label"])
runs = {}
x0 = torch.randn((B, 4, size // 8, size // 8), device=dev) * 14
for dual in (False, True):
    S.DUAL_STREAM = dual
    gs = S._GraphedSteps(model, c, uc, B, (size // 8, size // 8), 5.0, sig)
    gs.x.copy_(x0)
    for i in range(NS): gs._capture(i)
    runs[dual] = gs
torch.cuda.synchronize()
outs = {}
for rep in range(3):
    for dual, gs in runs.items():
        gs.x.copy_(x0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(NS): gs.graphs[i].replay()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / NS * 1e3
        outs[dual] = gs.x.clone()
        print(f"rep {rep} two_streams={dual}: {dt:.2f} ms/step", flush=True)
d = (outs[True] - outs[False]).double()
print("latent after %d steps: rel rms diff two-stream vs one-stream %.3e" % (NS, d.pow(2).mean().sqrt().item() / outs[False].double().pow(2).mean().sqrt().item()))
