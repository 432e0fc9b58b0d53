"""wall time of the phases of predict() for one batch of 4 at 512x512 alone on the chip (conditioning, sampling, decode)
and of predict_many over 6 batches — how much of the throughput-mode time is outside the sampling loops"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import config as C, pipeline, synth
dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
model = pipeline.build_engine(dev)
sampler = pipeline.init_sampling(50, 5.0, dev)
cfgs = C.default_runtime_config(steps=50, batch_size=4, noise_iters=0, gpu=0)
mk = lambda i: synth.synthetic_batch(4, 512, 512, 9, seed=i)


def t():
    torch.cuda.synchronize()
    return time.perf_counter()


for rep in range(3):
    b, buc = pipeline.prepare_batch(mk(rep), dev)
    t0 = t()
    c, uc = model.conditioner.get_unconditional_conditioning(b, batch_uc=buc, force_uc_zero_embeddings=cfgs.force_uc_zero_embeddings)
    x = sampler.get_init_noise(cfgs, model, cond=c, batch=b, uc=uc)
    t1 = t()
    z = sampler(model, x, cond=c, batch=b, uc=uc, init_step=0)
    t2 = t()
    img = torch.clamp((model.decode_first_stage(z) + 1.0) / 2.0, min=0.0, max=1.0)
    t3 = t()
    print(f"alone: conditioning {1e3*(t1-t0):.1f} ms, sampling {1e3*(t2-t1):.1f} ms, decode {1e3*(t3-t2):.1f} ms", flush=True)
for rep in range(3):
    bs = [mk(10 + i) for i in range(6)]
    t0 = t()
    pipeline.predict_many(cfgs, model, sampler, bs, dev)
    t1 = t()
    print(f"predict_many 6 batches: {1e3*(t1-t0):.1f} ms = {1e3*(t1-t0)/6:.1f} per batch", flush=True)
# sampling only, 3 in flight
xs, cs, ucs = [], [], []
for i in range(3):
    b, buc = pipeline.prepare_batch(mk(20 + i), dev)
    c, uc = model.conditioner.get_unconditional_conditioning(b, batch_uc=buc, force_uc_zero_embeddings=cfgs.force_uc_zero_embeddings)
    xs.append(sampler.get_init_noise(cfgs, model, cond=c, batch=b, uc=uc)); cs.append(c); ucs.append(uc)
for rep in range(3):
    t0 = t()
    sampler.sample_in_flight(model, xs, cs, ucs)
    t1 = t()
    print(f"sampling only, 3 in flight: {1e3*(t1-t0):.1f} ms = {1e3*(t1-t0)/3:.1f} per batch", flush=True)
