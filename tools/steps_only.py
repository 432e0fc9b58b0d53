"""run N sampler steps (for profiling): python tools/steps_only.py [n] [skip_k]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import pipeline, synth, lib as L
from sgm.modules.diffusionmodules.sampling import _Stepper
dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
B, size = 4, 512
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
model = pipeline.build_engine(dev)
sampler = pipeline.init_sampling(50, 5.0, dev)
b = synth.synthetic_batch(B, size, size, 9, seed=0)
b = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
batch, buc = pipeline.prepare_batch(b, dev)
c, uc = model.conditioner.get_unconditional_conditioning(batch, batch_uc=buc, force_uc_zero_embeddings=["label"])
st = _Stepper(model, c, uc, B, (size // 8, size // 8), 5.0)
sig = sampler._host_sigmas()
x = torch.randn((B, 4, size // 8, size // 8), device=dev) * 14.0
if len(sys.argv) > 2:
    L.check(L.load().udt_debug_set(b"skip_k", int(sys.argv[2])), "dbg")
for i in range(n): st.step(x, sig[5 + i], sig[6 + i])
torch.cuda.synchronize()
