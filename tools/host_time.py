"""host-side enqueue time of one sampler step vs its GPU time"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import pipeline, synth, lib as L
from sgm.modules.diffusionmodules.sampling import _Stepper
dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
B, size = 4, 512
model = pipeline.build_engine(dev)
sampler = pipeline.init_sampling(50, 5.0, dev)
b = synth.synthetic_batch(B, size, size, 9, seed=0)
b = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
batch, buc = pipeline.prepare_batch(b, dev)
c, uc = model.conditioner.get_unconditional_conditioning(batch, batch_uc=buc, force_uc_zero_embeddings=["label"])
st = _Stepper(model, c, uc, B, (size // 8, size // 8), 5.0)
sig = sampler._host_sigmas()
x = torch.randn((B, 4, size // 8, size // 8), device=dev) * 14.0
for i in range(3): st.step(x, sig[5 + i], sig[6 + i])
torch.cuda.synchronize()
for skip in (0, 1):
    L.check(L.load().udt_debug_set(b"skip_k", skip), "dbg")
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(5): st.step(x, sig[5 + i], sig[6 + i])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"skip_k={skip}: host enqueue {1e3*(t1-t0)/5:.2f} ms/step, until GPU done {1e3*(t2-t0)/5:.2f} ms/step", flush=True)
