"""stage timings of one batch outside the sampling loop + per-launch trace of the VAE decode"""
import collections, os, re, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import config as C, lib as L, ops, pipeline, synth
dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
B, size = 4, 512
model = pipeline.build_engine(dev)
b = synth.synthetic_batch(B, size, size, 9, seed=1)
b = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r
def cond():
    batch, buc = pipeline.prepare_batch(b, dev)
    return model.conditioner.get_unconditional_conditioning(batch, batch_uc=buc, force_uc_zero_embeddings=["label"])
ms_c, _ = timed(cond)
z = torch.randn((B, 4, size // 8, size // 8), device=dev)
ms_d, _ = timed(lambda: model.decode_first_stage(z))
print(f"conditioner (LabelEncoder + mask + VAE encode): {ms_c:.2f} ms   VAE decode: {ms_d:.2f} ms", flush=True)
lib = L.load()
for name, fn in (("decode", lambda: model.decode_first_stage(z)), ("conditioner", cond)):
    ops.prof_reset(); lib.udt_prof_trace(1); ops.prof_enable(0x3f)
    fn(); torch.cuda.synchronize()
    ops.prof_enable(0)
    out = os.path.join("gpurun_out", f"trace_{name}.csv")
    lib.udt_prof_dump(out.encode())
    agg = collections.OrderedDict(); tot = 0.0
    for line in open(out).read().splitlines()[1:]:
        cls, ms, tag = line.split(",", 2)
        ms = float(ms); tot += ms
        a = agg.setdefault(tag if tag else f"class{cls}", [0, 0.0]); a[0] += 1; a[1] += ms
    print(f"== {name}: traced {tot:.3f} ms")
    for tag, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f"{ms:8.3f} ms  x{n:3d}  {ms/n*1e3:8.1f} us  {tag}")
