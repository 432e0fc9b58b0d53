"""Round 6: one attend-and-excite gradient (B = 1, 512 x 512) or one training-step gradient (B = 4) a few times, eager launches, for
rocprofv3 --kernel-trace --stats (which kernels the reverse pass spends its time in).  usage: prof_aae.py [aae|train] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from udifftext_amd import backward, pipeline, synth, training as tr
mode = sys.argv[1] if len(sys.argv) > 1 else "aae"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0); torch.set_grad_enabled(False)
engine = pipeline.build_engine(dev)
unet = engine.model.diffusion_model
size = 512
h = size // 8
if mode == "aae":
    batch = synth.synthetic_batch(1, size, size, 9, seed=3)
    torch.manual_seed(1)
    batch, buc = pipeline.prepare_batch(batch, dev)
    c, uc = engine.conditioner.get_unconditional_conditioning(batch, batch_uc=buc, force_uc_zero_embeddings=["label"])
    x = torch.randn((1, 4, h, h), device=dev) * 5.0
    ts = torch.full((1,), 500.0, device=dev)
    f = lambda: backward.unet_local_loss_grad(unet, engine.loss_fn, x, ts, c["concat"], c["t_crossattn"], batch["mask"], batch["seg_mask"])
else:
    B = 4
    z = torch.randn((B, 4, h, h), device=dev)
    batch = synth.synthetic_batch(B, size, size, 9, seed=4)
    torch.manual_seed(2)
    batch, buc = pipeline.prepare_batch(batch, dev)
    cond = engine.conditioner(batch)
    seg = torch.zeros((B, 12, size, size), device=dev); seg[:, :9, size // 2 - 16:size // 2 + 16, :] = 1.0
    idx = torch.tensor([100, 400, 700, 900][:B])
    noise = torch.randn((B, 4, h, h), device=dev)
    f = lambda: tr.training_loss_and_grads(engine, z, cond, seg, batch["seg_mask"], sigma_idx=idx, noise=noise)
for _ in range(reps):
    f()
torch.cuda.synchronize()
