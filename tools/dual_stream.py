"""potential of two concurrent launch streams: two batches replayed as hipGraphs on two streams (each planned for half
the CUs) against one batch at a time on the whole GPU"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import pipeline, synth, lib as L
import sgm.modules.diffusionmodules.sampling as S
dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
B, size, NS = int(os.environ.get("B", 4)), 512, 8
model = pipeline.build_engine(dev)
sampler = pipeline.init_sampling(50, 5.0, dev)
sig = sampler._host_sigmas()
def cond_of(seed):
    b = synth.synthetic_batch(B, size, size, 9, seed=seed)
    b = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    batch, buc = pipeline.prepare_batch(b, dev)
    return model.conditioner.get_unconditional_conditioning(batch, batch_uc=buc, force_uc_zero_embeddings=["label"])
def build(seed, share, stream):
    L.check(L.load().udt_debug_set(b"cu_share", share), "dbg")
    c, uc = cond_of(seed)
    with torch.cuda.stream(stream):
        gs = S._GraphedSteps(model, c, uc, B, (size // 8, size // 8), 5.0, sig)
        gs.x.copy_(torch.randn_like(gs.x) * 14)
        for i in range(NS):
            gs._capture(i)
    torch.cuda.synchronize()
    return gs
S.DUAL_STREAM = False
streams = [torch.cuda.Stream() for _ in range(4)]
full = build(1, 1, streams[0])
def run_single(gs, stream):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for i in range(NS): gs.graphs[i].replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / NS * 1e3
def run_multi(sets):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(NS):
        for gs, st in zip(sets, streams):
            with torch.cuda.stream(st): gs.graphs[i].replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / NS * 1e3
print(f"whole GPU, one batch at a time: {run_single(full, streams[0]):.2f} {run_single(full, streams[0]):.2f} ms/step", flush=True)
sets = [build(10 + k, 2, streams[k]) for k in range(2)]
for rep in range(2):
    d = run_multi(sets)
    print(f"B={B}: 2 batches in flight: {d:.2f} ms per round = {d/2:.2f} ms/step/batch = {d/2/B:.3f} ms/step/image", flush=True)
