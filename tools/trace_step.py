"""Per-launch trace of ONE UNet sampler step at BASELINE config #2 (8 samples), aggregated by shape."""
import collections, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import config as C, lib as L, ops, pipeline, synth

dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
B, size = int(os.environ.get("B", 4)), int(os.environ.get("SIZE", 512))
model = pipeline.build_engine(dev)
sampler = pipeline.init_sampling(50, 5.0, dev)
batch, buc = pipeline.prepare_batch(synth.synthetic_batch(B, size, size, 9, seed=1), dev)
c, uc = model.conditioner.get_unconditional_conditioning(batch, batch_uc=buc, force_uc_zero_embeddings=["label"])
from sgm.modules.diffusionmodules.sampling import _Stepper
st = _Stepper(model, c, uc, B, (size // 8, size // 8), 5.0)
x = torch.randn((B, 4, size // 8, size // 8), device=dev) * 14
sig = sampler._host_sigmas()
for i in range(3):
    st.step(x, sig[i], sig[i + 1])
torch.cuda.synchronize()
lib = L.load()
ops.prof_reset(); lib.udt_prof_trace(1); ops.prof_enable(0x3f)
st.step(x, sig[3], sig[4])
torch.cuda.synchronize()
ops.prof_enable(0)
out = os.path.join("gpurun_out", os.environ.get("TRACE_OUT", "trace_step.csv"))
lib.udt_prof_dump(out.encode())
agg = collections.OrderedDict()
tot = 0.0
for line in open(out).read().splitlines()[1:]:
    cls, ms, tag = line.split(",", 2)
    ms = float(ms); tot += ms
    a = agg.setdefault(tag if tag else f"class{cls}", [0, 0.0])
    a[0] += 1; a[1] += ms
print(f"total traced {tot:.3f} ms")
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
for tag, (n, ms) in rows[:70]:
    tf = ""
    m = re.match(r"(?:gemm8?|lean\d|conv3p\S*) M=(\d+) N=(\d+) K=(\d+)", tag)
    if m:
        M, N, K = map(int, m.groups()); tf = f"{2.0*M*N*K*n/ms/1e9:7.0f} TF/s"
    m = re.match(r"attn B=(\d+) H=(\d+) nq=(\d+) nk=(\d+)", tag)
    if m:
        b, h, nq, nk = map(int, m.groups()); tf = f"{4.0*b*h*nq*nk*64*n/ms/1e9:7.0f} TF/s"
    print(f"{ms:8.3f} ms  x{n:3d}  {ms/n*1e3:8.1f} us  {tf:>12s}  {tag}")
