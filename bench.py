#!/usr/bin/env python
"""Benchmark of the UDiffText denoising hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input: BASELINE config #2 —
512x512, 50 deterministic Euler (== DDIM eta 0) steps with CFG 5.0, batch 4 per GPU, 9-character labels,
noise_iters 0 — i.e. conditioner (LabelEncoder + mask rescale + VAE encode of the masked image + 2 posterior
samples), 50 UNet calls on the CFG pair (8 samples), VAE decode, clamp.  Weights are the deterministic synthetic
recipe (no checkpoints exist here), inputs are seeded synthetic batches already resident in HBM when the
timed region starts.  For N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL) every rank
samples its own 4 images (weak scaling) and the decoded frames are all-gathered once per step.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel class (3x3 implicit-GEMM convolution):
algorithmic FLOPs of its launches / their summed duration, measured live with HIP events on the launch stream
(udt_prof_*) — on one extra eager pass of the same batch right after the timed region, because the timed region
replays hipGraphs (no per-launch host calls to bracket).  `cpu_baseline` times the CPU oracle (oracle/, a port pinned against the real
reference) on the host cores for a bounded sample and extrapolates (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_IMAGE = 83505e9       # 50 x 1597.326 + 1116.7 + 2514.6 + 7.2 GFLOP
PEAK_BF16 = 2500e12            # dense MFMA peak, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU (BASELINE config #2: 4)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--sampler-steps", type=int, default=50)
    ap.add_argument("--chars", type=int, default=9)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mode-table", action="store_true", help="skip the extra passes in the stricter launch modes")
    ap.add_argument("--in-flight", type=int, default=2, help="launch streams sampling concurrently per GPU (1 = one at a time)")
    ap.add_argument("--fuse", type=int, default=0, help="batches concatenated into one sampling batch per stream "
                    "(0 = automatic: the timed steps are spread over the streams, at most 4 per sampling batch)")
    return ap.parse_args()


def measured_traffic():
    """HBM bytes per launch of c3p::conv3p_kernel (the dominant kernel) over one batch of this workload, from the
    committed rocprofv3 PMC passes (profiles/r01_traffic.json, tools/collect_profiles.sh + tools/pmc_extrapolate.py:
    FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate passes).  Counter collection cannot run inside the
    timed region, hence the file; None if absent."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if not os.path.exists(path):
        return None
    try:
        return float(json.load(open(path))["conv3p"]["hbm_bytes_per_launch"])
    except Exception:
        return None


def cpu_baseline(model, size: int, chars: int, sampler_steps: int) -> dict:
    """time the CPU oracle on a bounded sample of the same workload: 2 UNet calls on one CFG pair at the bench
    resolution + LabelEncoder + 1 VAE encode + 1 VAE decode; extrapolate to sampler_steps UNet calls per image"""
    from oracle import nets, sampling, spec
    from udifftext_amd import synth
    cfg = spec.EngineConfig()
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    cores = torch.get_num_threads()
    batch = synth.synthetic_batch(1, size, size, chars, seed=0)
    h = size // 8
    with torch.no_grad():
        t0 = time.time(); ctx = nets.label_encoder(sd, batch["label"], cfg.label); t_label = time.time() - t0
        t0 = time.time(); mom = nets.vae_encode_moments(sd, batch["masked"], cfg.vae, "conditioner.embedders.2.model."); t_enc = time.time() - t0
        xin = torch.randn(2, 9, h, h)
        tctx = torch.cat([torch.zeros_like(ctx), ctx])
        ts = torch.tensor([999, 999])
        nets.unet_forward(sd, xin, ts, tctx, cfg.unet)                      # warm
        t0 = time.time()
        for _ in range(2):
            nets.unet_forward(sd, xin, ts, tctx, cfg.unet)
        t_unet = (time.time() - t0) / 2
        t0 = time.time(); nets.vae_decode(sd, torch.randn(1, 4, h, h), cfg.vae); t_dec = time.time() - t0
    per_image = sampler_steps * t_unet + t_enc + t_dec + t_label
    return {"value": 1.0 / per_image, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle (fp32 torch CPU): 2 warm UNet calls on one CFG pair @{size}x{size} ({t_unet:.2f} s each), "
                      f"1 VAE encode ({t_enc:.2f} s), 1 VAE decode ({t_dec:.2f} s), LabelEncoder ({t_label:.2f} s); "
                      f"extrapolated to {sampler_steps} UNet calls per image"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import udifftext_amd  # noqa: F401
    from udifftext_amd import config as C, lib as L, ops, pipeline, synth
    from udifftext_amd.parallel import gather_frames
    import sgm.modules.hipnn as H

    torch.set_grad_enabled(False)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):          # the conditioner announces its embedders like the reference does;
        model = pipeline.build_engine(dev)                # stdout carries the ONE JSON line only
    sampler = pipeline.init_sampling(args.sampler_steps, 5.0, dev)
    cfgs = C.default_runtime_config(steps=args.sampler_steps, batch_size=args.batch, noise_iters=0, gpu=local_rank)

    def make_batch(i):
        b = synth.synthetic_batch(args.batch, args.size, args.size, args.chars, seed=1000 * rank + i)
        return {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step(batch):
        samples, _ = pipeline.predict(cfgs, model, sampler, batch, dev)
        return gather_frames(samples, dist)

    def run_steps(blist, in_flight=None, fuse=None):
        """K steps = K batches; up to --in-flight of them are sampled concurrently on separate launch streams"""
        outs = pipeline.predict_many(cfgs, model, sampler, blist, dev, in_flight=args.in_flight if in_flight is None else in_flight,
                                     fuse=args.fuse if fuse is None else fuse)
        return [gather_frames(smp, dist) for smp, _ in outs]

    torch.manual_seed(1234 + rank)
    batches = [make_batch(i) for i in range(args.warmup + args.steps)]
    if args.warmup > 0:
        # untimed: W batches, topped up to the number of timed steps so that the same grouping (full groups in flight
        # plus whatever is left over) has captured its hipGraphs before the clock starts
        warm = [batches[i % args.warmup] for i in range(max(args.warmup, args.steps))]
        run_steps(warm)

    # ---- timed region ------------------------------------------------------------------------------------
    # (the sampling loop replays hipGraphs captured during warm-up, or on the first timed step when --warmup 0)
    barrier()
    t0 = time.perf_counter()
    frames = None
    frames = run_steps(batches[args.warmup:args.warmup + args.steps])[-1]
    barrier()
    elapsed = time.perf_counter() - t0

    # ---- roofline pass: graph replay issues no per-launch host calls, so the per-kernel HIP events (udt_prof_*,
    # recorded on the launch stream around every 3x3-convolution launch) are taken on ONE extra pass of the same
    # workload with eager launches, right after the timed region
    import sgm.modules.diffusionmodules.sampling as S
    graphs_on = bool(getattr(sampler, "use_graphs", False))
    sampler.use_graphs = False
    dual_prev, S.DUAL_STREAM = S.DUAL_STREAM, False     # one launch stream: every kernel is timed alone on the chip
    H.FLOP_COUNTER = {}
    ops.prof_reset()
    ops.prof_enable(1 << L.PROF_CONV3X3)
    one_step(batches[-1])
    torch.cuda.synchronize()
    ops.prof_enable(0)
    conv_ms, conv_launches = ops.prof_get(L.PROF_CONV3X3)
    conv_flops = H.FLOP_COUNTER.get("conv3x3", 0.0)
    conv_bytes = H.FLOP_COUNTER.get("conv3x3_bytes", 0.0)
    c3p_bytes, c3p_launches = H.FLOP_COUNTER.get("conv3p_bytes", 0.0), H.FLOP_COUNTER.get("conv3p_launches", 0.0)
    H.FLOP_COUNTER = None
    sampler.use_graphs = graphs_on
    S.DUAL_STREAM = dual_prev

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # ---- the same K steps in the stricter launch modes, for reference next to `value` (same barriers / max over ranks)
    def timed_mode(in_flight, fuse):
        blist = batches[args.warmup:args.warmup + args.steps]
        if args.warmup > 0:
            run_steps(blist, in_flight, fuse)                    # captures this mode's hipGraphs
        barrier()
        t1 = time.perf_counter()
        run_steps(blist, in_flight, fuse)
        barrier()
        tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return args.steps * args.batch * world / float(tt.item())

    other_modes = {}
    if not args.no_mode_table:
        if args.in_flight > 1 or args.fuse != 1:
            other_modes["one_batch_at_a_time"] = timed_mode(1, 1)
        if args.in_flight > 1 and args.fuse != 1:
            other_modes[f"{args.in_flight}_batches_in_flight_unfused"] = timed_mode(args.in_flight, 1)
    assert frames.shape[0] == args.batch * world and bool(torch.isfinite(frames).all())

    # per-step UNet time (second half of BASELINE's metric): ONE batch alone on the GPU, one sampler step on its CFG
    # pair, hipGraph replay (eager launches if graphs are off), averaged over 10 steps
    if rank == 0:
        n_meas = 10
        batch, buc = pipeline.prepare_batch(batches[0], dev)
        c, uc = model.conditioner.get_unconditional_conditioning(batch, batch_uc=buc, force_uc_zero_embeddings=["label"])
        sig = sampler._host_sigmas()
        hw = (args.size // 8, args.size // 8)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if graphs_on:
            gs = S._GraphedSteps(model, c, uc, args.batch, hw, 5.0, sig)
            gs.x.copy_(torch.randn_like(gs.x) * 14.0)
            for i in range(3, 3 + n_meas):
                gs._capture(i)
            with torch.cuda.stream(gs.capture_stream):
                gs.graphs[3].replay()
                e0.record()
                for i in range(3, 3 + n_meas):
                    gs.graphs[i].replay()
                e1.record()
        else:
            st = S._Stepper(model, c, uc, args.batch, hw, 5.0)
            x = torch.randn((args.batch, 4) + hw, device=dev) * 14.0
            for i in range(3):
                st.step(x, sig[i], sig[i + 1])
            torch.cuda.synchronize()
            e0.record()
            for i in range(n_meas):
                st.step(x, sig[3 + i], sig[4 + i])
            e1.record()
        torch.cuda.synchronize()
        unet_ms = e0.elapsed_time(e1) / n_meas

    fuse_eff = args.fuse if args.fuse > 0 else min(4, max(1, -(-args.steps // max(args.in_flight, 1))))
    if rank == 0:
        images = args.steps * args.batch * world
        value = images / elapsed
        achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        line = {
            "metric": f"{args.size}x{args.size} {args.sampler_steps}-step denoised images/sec (UDiffText hot path: conditioner + "
                      f"{args.sampler_steps} CFG Euler steps + VAE decode)",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "images_per_s_by_launch_mode": dict(other_modes, **{"throughput_mode (value)": value}),
            "unet_ms_per_sampler_step": unet_ms,
            "unet_ms_note": "one batch alone on the whole GPU (latency); with batches in flight the per-batch cost is lower",
            "config": {"workload": f"{args.size}x{args.size}, {args.sampler_steps} Euler/DDIM(eta 0) steps, CFG 5.0, "
                                   f"batch {args.batch} per step ({2 * args.batch * fuse_eff} samples per UNet call), {args.chars}-char "
                                   "labels, noise_iters 0; " + ("BASELINE.json configs[1]" if (args.size, args.batch, args.chars,
                                   args.sampler_steps) == (512, 4, 9, 50) else "BASELINE.json configs[3]" if (args.size, args.batch,
                                   args.chars) == (768, 8, 12) else "non-baseline shape"),
                       "global_batch": args.batch * world, "parallelism": f"dp{world} (images sharded, one all-gather of frames)",
                       "weights": "synthetic (name-keyed recipe), 1361.2 M parameters",
                       "launch": "hipGraph replay of the 50 sampler steps" if graphs_on else "eager kernel launches",
                       "in_flight": (f"throughput mode of pipeline.predict_many: {fuse_eff} consecutive batches concatenated per "
                                     f"sampling batch, {args.in_flight} sampling batches concurrently per GPU (one launch stream "
                                     f"each, planned for 1/{args.in_flight} of the CUs); left-overs run in smaller groups")
                                    if (args.in_flight > 1 or fuse_eff > 1) else "one batch at a time"},
            "roofline": {"kernel": "3x3 convolution: c3p::conv3p_kernel (LDS-staged patches) + g8::gemm8_kernel<CONV> "
                                   "(stride-2 / upsampling gathers), UNet + VAE", "bound": "mfma",
                         "achieved": achieved, "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s", "frac": achieved / (PEAK_BF16 / 1e12),
                         "traffic": measured_traffic(), "launches": conv_launches,
                         "measured_on": "one eager single-stream pass of the same batch right after the timed region: HIP "
                                        "events around every launch, each kernel alone on the chip (the timed region "
                                        "replays hipGraphs with batches in flight, where launches of two streams overlap)",
                         "avg_launch_us": conv_ms * 1e3 / max(conv_launches, 1),
                         "algorithmic_gflop_per_launch": conv_flops / max(conv_launches, 1) / 1e9,
                         "algorithmic_bytes_per_launch": conv_bytes / max(conv_launches, 1),
                         "traffic_scope": "c3p::conv3p_kernel launches only (%d of the %d launches of the class): algorithmic "
                                          "%.1f MB per launch" % (int(c3p_launches), conv_launches, c3p_bytes / max(c3p_launches, 1) / 1e6),
                         "whole_path_frac_of_peak": value * FLOP_PER_IMAGE / (world * PEAK_BF16)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(model, args.size, args.chars, args.sampler_steps)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
