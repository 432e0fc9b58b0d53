"""Round 6: time of one attend-and-excite gradient (tape-mode forward + reverse pass through the whole UNet, B = 1) and
of one training step's loss + parameter gradients (B = 4) at 64 x 64 latents (512 x 512 images), with the per-class split from the
library's launch profiler.  Correctness-first kernels: the numbers say where this path stands, not where it could be."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import udifftext_amd
from udifftext_amd import backward, pipeline, synth, training as tr, ops, lib as L
dev = torch.device("cuda", 0); torch.set_grad_enabled(False)
engine = pipeline.build_engine(dev)
unet = engine.model.diffusion_model

def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for size, B in ((512, 1), (256, 1)):
    batch = synth.synthetic_batch(B, size, size, 9, seed=3)
    torch.manual_seed(1)
    batch, buc = pipeline.prepare_batch(batch, dev)
    c, uc = engine.conditioner.get_unconditional_conditioning(batch, batch_uc=buc, force_uc_zero_embeddings=["label"])
    h = size // 8
    x = torch.randn((B, 4, h, h), device=dev) * 5.0
    ts = torch.full((B,), 500.0, device=dev)
    f = lambda: backward.unet_local_loss_grad(unet, engine.loss_fn, x, ts, c["concat"], c["t_crossattn"], batch["mask"], batch["seg_mask"])
    ms = timed(f)
    sampler = pipeline.init_sampling(50, 5.0, dev)
    print(f"attend-and-excite gradient, {size}x{size}, B={B}: {ms:.1f} ms per evaluation (tape-mode forward + reverse pass, eager launches)")
    args = (ts, c["concat"], c["t_crossattn"], batch["mask"], batch["seg_mask"])
    runner = backward.GraphedLocalLossGrad(unet, engine.loss_fn, x, *args)
    msg = timed(lambda: runner(x, *args), n=10)
    print(f"attend-and-excite gradient, {size}x{size}, B={B}: {msg:.1f} ms per evaluation replayed as one hipGraph")
    del runner
for size, B in ((512, 4), (256, 4)):
    h = size // 8
    z = torch.randn((B, 4, h, h), device=dev)
    batch = synth.synthetic_batch(B, size, size, 9, seed=4)
    torch.manual_seed(2)
    batch, buc = pipeline.prepare_batch(batch, dev)
    cond = engine.conditioner(batch)
    seg = torch.zeros((B, 12, size, size), device=dev); seg[:, :9, size // 2 - 16:size // 2 + 16, :] = 1.0
    idx = torch.tensor([100, 400, 700, 900][:B])
    noise = torch.randn((B, 4, h, h), device=dev)
    f = lambda: tr.training_loss_and_grads(engine, z, cond, seg, batch["seg_mask"], sigma_idx=idx, noise=noise)
    ms = timed(f)
    f2 = lambda: tr.training_loss_and_grads(engine, z, cond, seg, batch["seg_mask"], sigma_idx=idx, noise=noise, want_grads=False)
    ms2 = timed(f2)
    print(f"training step loss + gradients of the 112 t_attn / t_norm tensors, {size}x{size}, B={B}: {ms:.1f} ms (tape-mode forward alone {ms2:.1f} ms; "
          f"inference forward of the same 4 samples: see unet_ms_per_sampler_step / 2)")

# end to end: predict (test.py:19-40) of ONE 512 x 512 image, 50 steps, with and without attend-and-excite
from udifftext_amd import config as C
for aae in (False, True):
    sampler = pipeline.init_sampling(50, 5.0, dev)
    cfgs = C.default_runtime_config(steps=50, batch_size=1, noise_iters=0)
    cfgs.aae_enabled = aae
    ts = []
    for rep in range(2):
        batch = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in synth.synthetic_batch(1, 512, 512, 9, seed=30 + rep).items()}
        torch.manual_seed(7 + rep)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            frames, z = pipeline.predict(cfgs, engine, sampler, batch, dev)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    extra = ""
    if aae:
        st = getattr(sampler, "last_aae_stats", None)
        extra = f" ({st})" if st else ""
    print(f"predict, one 512x512 image, 50 steps, aae_enabled={aae}: {ts[-1]:.2f} s (first call {ts[0]:.2f} s){extra}")
