"""The reference's DEFAULT workload (configs/test.yaml: batch_size 1, noise_iters 10, 50 steps) — seconds per image through
pipeline.predict, noise search included.   python tools/bench_reference_default.py   (UDT_NOISE_BATCH=0|1)"""
import contextlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import config as C, pipeline, synth

dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
model = pipeline.build_engine(dev)
sampler = pipeline.init_sampling(50, 5.0, dev)
cfg = C.default_runtime_config(steps=50, batch_size=1, noise_iters=10, gpu=0)
batches = [synth.synthetic_batch(1, 512, 512, 9, seed=10 + i) for i in range(5)]
with contextlib.redirect_stdout(sys.stderr):
    for b in batches[:2]:
        pipeline.predict(cfg, model, sampler, dict(b), dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches[2:]:
        pipeline.predict(cfg, model, sampler, dict(b), dev)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"UDT_NOISE_BATCH={os.environ.get('UDT_NOISE_BATCH', '1')}: {dt:.4f} s per image = {1 / dt:.3f} images/s (batch 1, noise_iters 10, 50 steps)")
