"""Timed CPU baseline of bench.py (``cpu_baseline``): the oracle on the host cores, in a PROCESS OF ITS OWN.

Round 4's baseline ran inside the benchmark process (HIP runtime helper threads alive, torch's default of one intra-op thread
per LOGICAL core, unbound) and gave 52 s on one lease and 190 s on the driver's for the same work.  Here the parent sets
OMP_NUM_THREADS to the physical cores, OMP_PROC_BIND=close / OMP_PLACES=cores before this interpreter starts, nothing else
runs in the process, the UNet call (99.5 % of the per-image time) and the LabelEncoder have a warm run in front of them, and the
timed UNet call is repeated in two rounds whose times are both reported (``unet_call_s``): the same box should now give the same
number twice.  The VAE encode / decode are timed ONCE, cold (first-touch allocation and primitive creation included): together
they are < 1 % of the extrapolated per-image time, a warm run of each would cost more host seconds than it corrects.

Sample (bounded, ~10-30 s of CPU work on the GPU box's host): at the bench resolution — LabelEncoder, 1 VAE encode (cold), 1 warm-up
+ 2 timed UNet calls on one CFG pair (round 6: two rounds of ONE call, was two of two — the bench command also runs configs #5 / #4
now), 1 VAE decode (cold); extrapolated to ``sampler_steps`` UNet calls per image (every step costs the same).
``--config1``: additionally BASELINE config #1 in full (256x256, 10 steps, 4 characters: ~1-3 minutes).

Weights: the name-keyed synthetic recipe (udifftext_amd/synth.py) generated here from oracle.spec's shape list — the same
tensors the GPU engine holds (tests/test_oracle_golden.py pins both against the reference's state dict).
Prints ONE JSON object.  Test infrastructure: only bench.py's cpu_baseline leg executes this module.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--chars", type=int, default=9)
    ap.add_argument("--sampler-steps", type=int, default=50)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--config1", action="store_true")
    args = ap.parse_args()
    if args.threads > 0:
        torch.set_num_threads(args.threads)
    from oracle import nets, sampling, spec
    from udifftext_amd import synth
    cfg = spec.EngineConfig()
    t0 = time.time()
    sd = synth.synthetic_state_dict(spec.engine_param_shapes(cfg))
    sd["denoiser.sigmas"] = sampling.denoiser_sigma_table(1000)                       # (the engine's computed buffers)
    sd["loss_fn.g_kernel"] = sampling.gaussian_kernel(3, 1.0, 12)
    sd["conditioner.embedders.0.pos_embedding.pe"] = nets.positional_encoding(12, 2048)
    t_weights = time.time() - t0
    size, h = args.size, args.size // 8
    batch = synth.synthetic_batch(1, size, size, args.chars, seed=0)
    load0 = os.getloadavg()[0]
    with torch.no_grad():
        torch.manual_seed(0)
        nets.label_encoder(sd, batch["label"], cfg.label)                                             # warm
        t0 = time.time(); ctx = nets.label_encoder(sd, batch["label"], cfg.label); t_label = time.time() - t0
        xin = torch.randn(2, 9, h, h)
        tctx = torch.cat([torch.zeros_like(ctx), ctx])
        ts = torch.tensor([999, 999])
        nets.unet_forward(sd, xin, ts, tctx, cfg.unet)                                                # warm
        rounds = []
        for _ in range(2):
            t0 = time.time()
            nets.unet_forward(sd, xin, ts, tctx, cfg.unet)
            rounds.append(time.time() - t0)
        t_unet = min(rounds)
        pre = "conditioner.embedders.2.model." if any(k.startswith("conditioner.embedders.2.model.") for k in sd) else "first_stage_model."
        t0 = time.time(); nets.vae_encode_moments(sd, batch["masked"], cfg.vae, pre); t_enc = time.time() - t0
        t0 = time.time(); nets.vae_decode(sd, torch.randn(1, 4, h, h), cfg.vae); t_dec = time.time() - t0
        t_c1 = None
        if args.config1:
            t0 = time.time()
            sampling.predict(sd, cfg, synth.synthetic_batch(1, 256, 256, 4, seed=0), steps=10, scale=5.0)
            t_c1 = time.time() - t0
    per_image = args.sampler_steps * t_unet + t_enc + t_dec + t_label
    out = {"value": 1.0 / per_image, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
           "threads": torch.get_num_threads(), "logical_cpus": os.cpu_count(), "omp_proc_bind": os.environ.get("OMP_PROC_BIND", ""),
           "load_average_before": round(load0, 2), "load_average_after": round(os.getloadavg()[0], 2),
           "unet_call_s": [round(r, 4) for r in rounds], "vae_encode_s": round(t_enc, 4), "vae_decode_s": round(t_dec, 4),
           "label_encoder_s": round(t_label, 4), "weights_generated_s": round(t_weights, 1),
           "sample": f"oracle (fp32 torch CPU, own process, {torch.get_num_threads()} threads bound close to cores): at {size}x{size}: "
                     f"2 warm UNet calls on one CFG pair, timed one by one ({rounds[0]:.2f} / {rounds[1]:.2f} s, the faster "
                     f"counts), 1 VAE encode ({t_enc:.2f} s, cold), 1 VAE decode ({t_dec:.2f} s, cold), LabelEncoder ({t_label:.2f} s, warm); "
                     f"extrapolated to {args.sampler_steps} UNet calls per image"}
    if t_c1 is not None:
        out["config1_full_run_s"] = t_c1
        out["config1_images_per_s"] = 1.0 / t_c1
    print(json.dumps(out))


if __name__ == "__main__":
    main()
