"""one predict() of the bench workload (conditioner + sampler steps + VAE decode), for counter collection"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import config as C, pipeline, synth
dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
model = pipeline.build_engine(dev)
sampler = pipeline.init_sampling(steps, 5.0, dev)
cfgs = C.default_runtime_config(steps=steps, batch_size=4, noise_iters=0, gpu=0)
b = synth.synthetic_batch(4, 512, 512, 9, seed=0)
b = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
print("stage: predict", flush=True)
# UDT_PLAN_SHARE=n: every launch is PLANNED as one of n batches in flight (udt_gemm_desc.cu_share: the share-aware split-K / tile
# rules of the headline mode) while this process still runs one batch alone — counter collection serialises kernels anyway
from udifftext_amd import ops
with ops.launch_context(cu_share=int(os.environ.get("UDT_PLAN_SHARE", "1"))):
    out, z = pipeline.predict(cfgs, model, sampler, b, dev)
torch.cuda.synchronize()
print("done", out.shape, flush=True)
