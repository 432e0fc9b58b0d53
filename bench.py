#!/usr/bin/env python
"""Benchmark of the UDiffText denoising hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W                        # BASELINE config #2 per GPU (weak scaling)
    python bench.py --size 768 --batch 8 --chars 12                      # BASELINE config #4
    torchrun ... bench.py --gpus 8 --global-batch 64                     # BASELINE config #3 (one sharded global batch)

One "step" = one pass of the hot path over one GLOBAL batch of synthetic input.  Default (BASELINE config #2):
512x512, 50 deterministic Euler (== DDIM eta 0) steps with CFG 5.0, batch 4 per GPU, 9-character labels,
noise_iters 0 — i.e. conditioner (LabelEncoder + mask rescale + VAE encode of the masked image + 2 posterior
samples), 50 UNet calls on the CFG pair (8 samples per call), VAE decode, clamp.  Weights are the deterministic
synthetic recipe (no checkpoints exist here), inputs are seeded synthetic batches already resident in HBM when the
timed region starts.  The global batch (--global-batch, default batch x N) is sharded over the N ranks by
``parallel.predict_sharded`` (per-image seeds: results do not depend on N) and the decoded frames are all-gathered
once per global batch over RCCL.

`value` is the on-config number: every UNet call runs on ONE batch of --batch images (its CFG pair); three such
batches are in flight per GPU on three launch streams (--in-flight 3, stated in config.workload).  The other launch
modes (one batch at a time; throughput mode with batches concatenated into larger UNet calls) are reported under
`images_per_s_by_launch_mode` and are never `value`.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel class (3x3 convolution); `roofline_classes`
adds the GEMM (linears + 1x1 convolutions) and flash-attention classes: algorithmic FLOPs of the launches / their
summed duration, measured live with HIP events on the launch stream (udt_prof_*) on one extra eager pass of the same
batch right after the timed region (the timed region replays hipGraphs: no per-launch host calls to bracket).
`cpu_baseline` times the CPU oracle (oracle/, a port pinned against the real reference) on the host cores for a
bounded sample and extrapolates (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import contextlib
import glob
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# UNet GFLOP per image-step / VAE encode / VAE decode / LabelEncoder, by image size
WORK = {256: (355.320, 272.7, 622.2), 512: (1597.326, 1116.7, 2514.6), 768: (4279.809, 2609.1, 5754.4)}
PEAK_BF16 = 2500e12            # dense MFMA peak, MI355X_MICROARCH.md
PEAK_HBM = 8.0e12


def flop_per_image(size: int, sampler_steps: int) -> float:
    unet, enc, dec = WORK.get(size, WORK[512])
    return (sampler_steps * unet + enc + dec + 7.2) * 1e9


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="images per UNet call and GPU (BASELINE config #2: 4)")
    ap.add_argument("--global-batch", type=int, default=0, help="images per step over ALL GPUs (0 = batch x GPUs); "
                    "config #3: 64 on 8 GPUs")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--sampler-steps", type=int, default=50)
    ap.add_argument("--chars", type=int, default=9)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-config1", action="store_true", help="the CPU baseline also runs BASELINE config #1 in full (256x256, 10 steps: "
                    "minutes of host time; off by default so that the bench command stays a few minutes long)")
    ap.add_argument("--no-mode-table", action="store_true", help="skip the extra passes in the other launch modes")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the short BASELINE config #5 (--fp8) and config #4 (768x768, "
                    "batch 8, 12 characters) passes that the default single-GPU config-#2 run appends under `config5` / `config4`")
    ap.add_argument("--in-flight", type=int, default=3, help="batches sampled concurrently per GPU, one launch stream each "
                    "(same-box on MI355X: 1 -> 5.7, 2 -> 7.13, 3 -> 7.67, 4 -> 6.0 images/s)")
    ap.add_argument("--fp8", action="store_true", help="BASELINE config #5 arithmetic: every linear of the 640- / 1280-channel "
                    "transformer blocks on MX8 operands (e4m3 + E8M0 block scales written by the producers' epilogues) and the "
                    "self-attention's Q K^T / P V on e4m3 operands (UDT_FP8_ATTN=0: bf16 attention)")
    ap.add_argument("--noise-iters", type=int, default=10, help="noise search iterations of the reference-default measurement "
                    "(configs/test.yaml:14 noise_iters: 10, batch_size 1): reported as images_per_s_reference_default")
    ap.add_argument("--no-reference-default", action="store_true", help="skip the reference-default (B = 1, noise search) pass")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port when bench.py launches its own ranks")
    ap.add_argument("--stub", action="store_true", help="host-logic self-test on the CPU (gloo, a stub engine): rank spawning, "
                    "image sharding, the single all-gather per global batch and the JSON line; the numbers mean nothing")
    ap.add_argument("--fuse", type=int, default=1, help="batches concatenated into one UNet call (1 = the on-config "
                    "batch per call; > 1 or 0 = throughput mode, never the headline value)")
    return ap.parse_args()


def _round_files(suffix: str):
    """profiles/rNN_<suffix> of the newest round first — the round's FINAL collection only (rNN_, not the earlier rNNa_ / rNNe_ ones)"""
    import re
    pat = re.compile(r"r(\d+)_" + re.escape(suffix) + "$")
    hits = [(int(m.group(1)), p) for p in glob.glob(os.path.join(ROOT, "profiles", "r*_" + suffix)) for m in [pat.match(os.path.basename(p))] if m]
    return [p for _, p in sorted(hits, reverse=True)]


def measured_traffic():
    """HBM bytes per launch of the patch-staged 3x3 convolution kernels (wd::wconv3_kernel + lg::lconv3_kernel: the dominant
    kernels) over one batch of this workload, from the committed rocprofv3 PMC passes (profiles/rNN_traffic.json, newest round; tools/collect_profiles.sh +
    tools/pmc_extrapolate.py: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate passes).  Counter collection
    cannot run inside the timed region, hence the file; None if absent."""
    for path in _round_files("traffic.json"):
        try:
            d = json.load(open(path))
            src = os.path.basename(path) + (" @ " + str(d["head"]) if d.get("head") else "")      # the commit it was collected at
            return float((d.get("conv3") or d["conv3p"])["hbm_bytes_per_launch"]), src
        except Exception:
            continue
    return None, None


def measured_traffic_inflight_plans():
    """the same counters with every launch PLANNED for three batches in flight (tools/predict_once.py UDT_PLAN_SHARE=3: the split-K /
    tile plans of the headline mode; profiles/rNN_traffic_inflight_plans.json) — None if absent"""
    for path in _round_files("traffic_inflight_plans.json"):
        try:
            d = json.load(open(path))
            return float((d.get("conv3") or d["conv3p"])["hbm_bytes_per_launch"]), os.path.basename(path) + (" @ " + str(d["head"]) if d.get("head") else "")
        except Exception:
            continue
    return None, None


def physical_cores() -> int:
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_baseline(size: int, chars: int, sampler_steps: int, config1: bool = False) -> dict:
    """time the CPU oracle on a bounded sample IN ITS OWN PROCESS (oracle/cpu_bench.py): one thread per physical core, bound
    (OMP_PROC_BIND=close, OMP_PLACES=cores — set before the child's OpenMP runtime starts), nothing else in the process, a warm run
    in front of the UNet call and the LabelEncoder (the VAE passes, < 1 % of the per-image time, are timed cold), the UNet call timed
    twice, both reported.  At the bench resolution: 1 + 2 UNet calls on one CFG pair, LabelEncoder, 1 VAE encode, 1 VAE decode, extrapolated to sampler_steps UNet calls per image (every
    step costs the same); --cpu-config1 adds BASELINE config #1 in full (minutes)."""
    import subprocess
    n = physical_cores()
    env = dict(os.environ)
    env.update({"OMP_NUM_THREADS": str(n), "MKL_NUM_THREADS": str(n), "OMP_PROC_BIND": "close", "OMP_PLACES": "cores",
                "HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py"), "--size", str(size), "--chars", str(chars),
           "--sampler-steps", str(sampler_steps), "--threads", str(n)] + (["--config1"] if config1 else [])
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"value": None, "unit": "images/s", "cores": n, "kind": "port", "sample": "oracle/cpu_bench.py failed: " + r.stderr[-300:]}
    return json.loads(lines[-1])


def extra_config(flags: list, steps: int, warmup: int, value_bf16: float | None = None) -> dict:
    """one more BASELINE config on this box, right after the main line's timed region: THIS script in a process of its own (the same
    code path as a direct `python bench.py <flags>` run: own engine, own hipGraphs, same barriers around exactly `steps` timed steps),
    reduced to the keys that identify the workload and its numbers.  Reported beside `value`, never as it."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline",
           "--no-reference-default", "--no-mode-table", "--no-extra-configs"] + flags
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "UDT_BENCH_FORCE_DIST")}
    t0 = time.perf_counter()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"value": None, "error": "bench.py " + " ".join(flags) + " failed: " + r.stderr[-300:]}
    d = json.loads(lines[-1])
    out = {k: d[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "unet_ms_per_sampler_step") if k in d}
    out["workload"] = d["config"]["workload"]
    out["flags"] = " ".join(flags)
    out["roofline_classes"] = dict({"conv3x3": {k: d["roofline"][k] for k in ("achieved", "frac", "launches", "avg_launch_us")}},
                                   **{n: {k: c[k] for k in ("achieved", "frac", "launches", "avg_launch_us")} for n, c in d["roofline_classes"].items()})
    out["whole_path_frac_of_peak"] = d["roofline"]["whole_path_frac_of_peak"]
    if d["roofline"].get("whole_path_frac_of_fp8_peak") is not None:
        out["whole_path_frac_of_fp8_peak"] = d["roofline"]["whole_path_frac_of_fp8_peak"]
    if value_bf16:
        out["ratio_to_value"] = d["value"] / value_bf16
        out["ratio_note"] = "this pass's images/s over the main line's bf16 `value`: same box, same steps / warm-up, back to back"
    out["wall_s"] = time.perf_counter() - t0
    return out


def self_launch(args) -> int:
    """spawn --gpus ranks of this script with torch.distributed.run's environment contract (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT); stdout of rank 0 (the ONE JSON line) is passed through"""
    import socket
    import subprocess
    port = args.master_port
    if port <= 0:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for pr in procs:
        rc = pr.wait() or rc
    return rc


def stub_main(args, world: int, rank: int) -> None:
    """--stub: the multi-rank plumbing of this script with a stub engine on the CPU (tests/test_bench_cpu.py)"""
    import torch.distributed as dist
    from udifftext_amd import config as C, parallel, rng

    class Cond:
        def get_unconditional_conditioning(self, batch, batch_uc=None, force_uc_zero_embeddings=None):
            B = batch["image"].shape[0]
            f = batch["image"].mean(dim=(1, 2, 3)).reshape(B, 1, 1, 1)
            return {"concat": rng.randn((B, 4, 2, 2)) + f}, {"concat": rng.randn((B, 4, 2, 2)) + f}

    class Model:
        conditioner = Cond()

        def decode_first_stage(self, z):
            return z[:, :3].repeat_interleave(2, -1).repeat_interleave(2, -2) * 0.1

    class Sampler:
        def get_init_noise(self, cfgs, model, cond, batch, uc=None):
            return rng.randn((cfgs.batch_size, 4, 2, 2))

        def sample_in_flight(self, model, xs, conds, ucs, init_step=0, deferred_checks=None, streams=None):
            return [x + 0.5 * c["concat"] - 0.25 * u["concat"] for x, c, u in zip(xs, conds, ucs)]

    on = world > 1
    if on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    G = args.global_batch if args.global_batch > 0 else args.batch * world
    cfgs = C.default_runtime_config(steps=2, batch_size=args.batch, noise_iters=0)

    def gb(i):
        g = torch.Generator().manual_seed(1000 + i)
        return {"image": torch.rand((G, 3, 8, 8), generator=g), "label": [f"i{k}" for k in range(G)], "txt": [""] * G,
                "name": [str(k) for k in range(G)], "target_size_as_tuple": torch.tensor([[4, 4]] * G)}

    batches = [gb(i) for i in range(args.warmup + args.steps)]
    seeds = [77 + i for i in range(len(batches))]
    run = lambda b, sd: parallel.predict_sharded(cfgs, Model(), Sampler(), b, sd, dist=dist if on else None, micro_batch=args.batch,
                                                 in_flight=args.in_flight, fuse=1, device=torch.device("cpu"))
    if args.warmup:
        run(batches[:args.warmup], seeds[:args.warmup])
    if on:
        dist.barrier()
    t0 = time.perf_counter()
    frames = run(batches[args.warmup:], seeds[args.warmup:])[-1]
    if on:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if on:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert frames.shape[0] == G
    if rank == 0:
        print(json.dumps({"metric": "stub (host logic only)", "value": args.steps * G / float(t.item()), "unit": "images/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(t.item()) / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak" if args.global_batch <= 0 else "strong", "vs_baseline": None,
                          "data": "stub", "dtype": "f32", "config": {"workload": "stub engine on the CPU", "global_batch": G}}), flush=True)
    if on:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torchrun: launch the N ranks ourselves (one process per GPU, RCCL rendezvous on 127.0.0.1) and relay
        # rank 0's JSON line — `python bench.py --gpus 8` and `python -m torch.distributed.run ... bench.py --gpus 8` are
        # the same measurement
        raise SystemExit(self_launch(args))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    if args.stub:
        return stub_main(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("UDT_BENCH_FORCE_DIST"):     # (the env switch exercises the RCCL path on a 1-GPU box)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(args.master_port or 29541))     # (set by torchrun; the forced one-rank world has none)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import udifftext_amd  # noqa: F401
    from udifftext_amd import config as C, lib as L, ops, parallel, pipeline, synth

    import sgm.modules.hipnn as Hnn
    if args.fp8:
        Hnn.FP8_LINEARS = True
    Hnn_fp8_attention = Hnn.fp8_attention
    torch.set_grad_enabled(False)
    with contextlib.redirect_stdout(sys.stderr):          # the conditioner announces its embedders like the reference does;
        model = pipeline.build_engine(dev)                # stdout carries the ONE JSON line only
    sampler = pipeline.init_sampling(args.sampler_steps, 5.0, dev)
    cfgs = C.default_runtime_config(steps=args.sampler_steps, batch_size=args.batch, noise_iters=0, gpu=local_rank)
    G = args.global_batch if args.global_batch > 0 else args.batch * world
    weak = args.global_batch <= 0

    def make_global_batch(i):
        # identical on every rank (same seed): each rank slices its shard out of it
        b = synth.synthetic_batch(G, args.size, args.size, args.chars, seed=1000 + i)
        return {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(gbs, seeds, in_flight=None, fuse=None):
        """K steps = K global batches; every rank samples its shard (micro-batches of --batch images, --in-flight of
        them concurrently) and joins ONE all-gather of frames per global batch"""
        return parallel.predict_sharded(cfgs, model, sampler, gbs, seeds, dist=dist, micro_batch=args.batch,
                                        in_flight=args.in_flight if in_flight is None else in_flight,
                                        fuse=args.fuse if fuse is None else fuse, device=dev)

    n_b = args.warmup + args.steps
    batches = [make_global_batch(i) for i in range(n_b)]
    seeds = [77 + i for i in range(n_b)]
    timed_b, timed_s = batches[args.warmup:], seeds[args.warmup:]
    if args.warmup > 0:
        # untimed: W batches, topped up to the number of timed steps so that the same grouping (full groups in flight
        # plus whatever is left over) has captured its hipGraphs before the clock starts
        idx = [i % args.warmup for i in range(max(args.warmup, args.steps))]
        run_steps([batches[i] for i in idx], [seeds[i] for i in idx])

    # ---- timed region ------------------------------------------------------------------------------------
    # (the sampling loop replays hipGraphs captured during warm-up, or on the first timed step when --warmup 0)
    barrier()
    t0 = time.perf_counter()
    frames = run_steps(timed_b, timed_s)[-1]
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    assert frames.shape[0] == G and bool(torch.isfinite(frames).all())

    # ---- roofline pass: graph replay issues no per-launch host calls, so the per-kernel HIP events (udt_prof_*,
    # recorded on the launch stream around every launch of the three MFMA classes) are taken on ONE extra pass of one
    # local batch with eager launches on a single stream, right after the timed region
    import sgm.modules.diffusionmodules.sampling as S
    graphs_on = bool(getattr(sampler, "use_graphs", False))
    sampler.use_graphs = False
    dual_prev, S.DUAL_STREAM = S.DUAL_STREAM, False     # one launch stream: every kernel is timed alone on the chip
    lo, hi = parallel.shard_range(G, rank, world)
    local = parallel.slice_batch(batches[-1], lo, min(hi, lo + args.batch))
    cfg1 = C.default_runtime_config(steps=args.sampler_steps, batch_size=len(local["label"]), noise_iters=0, gpu=local_rank)
    ops.WORK_COUNTER = {}
    ops.prof_reset()
    ops.prof_enable((1 << L.PROF_CONV3X3) | (1 << L.PROF_GEMM) | (1 << L.PROF_ATTN))
    pipeline.predict(cfg1, model, sampler, local, dev)
    torch.cuda.synchronize()
    ops.prof_enable(0)
    conv_ms, conv_launches = ops.prof_get(L.PROF_CONV3X3)
    gemm_ms, gemm_launches = ops.prof_get(L.PROF_GEMM)
    attn_ms, attn_launches = ops.prof_get(L.PROF_ATTN)
    W = ops.WORK_COUNTER
    ops.WORK_COUNTER = None
    sampler.use_graphs = graphs_on
    S.DUAL_STREAM = dual_prev
    conv_flops, conv_bytes = W.get("conv3x3", 0.0), W.get("conv3x3_bytes", 0.0)
    c3p_bytes, c3p_launches = W.get("conv3p_bytes", 0.0), W.get("conv3p_launches", 0.0)
    gemm_flops = W.get("gemm", 0.0) + W.get("conv1x1", 0.0) + W.get("gemm_fp8", 0.0)
    gemm_bytes = W.get("gemm_bytes", 0.0) + W.get("conv1x1_bytes", 0.0)
    attn_flops, attn_bytes = W.get("attn", 0.0) + W.get("attn_fp8", 0.0), W.get("attn_bytes", 0.0) + W.get("attn_fp8_bytes", 0.0)

    # ---- the same K steps in the other launch modes, for reference next to `value` (same barriers / max over ranks)
    def timed_mode(in_flight, fuse):
        if args.warmup > 0:
            run_steps(timed_b, timed_s, in_flight, fuse)         # captures this mode's hipGraphs
        barrier()
        t1 = time.perf_counter()
        run_steps(timed_b, timed_s, in_flight, fuse)
        barrier()
        tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return args.steps * G / float(tt.item())

    other_modes = {}
    if not args.no_mode_table:
        if not (args.in_flight == 1 and args.fuse == 1):
            other_modes["one_batch_at_a_time"] = timed_mode(1, 1)
        if args.in_flight > 2 and args.fuse == 1:
            other_modes["2_batches_in_flight"] = timed_mode(2, 1)
        if args.fuse == 1 and args.steps * (G // world) // args.batch >= 4:
            other_modes["throughput_mode_batches_concatenated_per_unet_call (off-config)"] = timed_mode(2, 0)

    # per-step UNet time (second half of BASELINE's metric): ONE batch alone on the GPU, one sampler step on its CFG
    # pair, hipGraph replay (eager launches if graphs are off), averaged over 10 steps
    if rank == 0:
        n_meas = 10
        batch, buc = pipeline.prepare_batch(parallel.slice_batch(batches[0], 0, args.batch), dev)
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

    # ---- the reference's DEFAULT workload (configs/test.yaml:14,22: batch_size 1, noise_iters 10 -> 2 x 10 + 50 = 70 UNet
    # calls per image on one CFG pair): one image at a time through pipeline.predict, noise search included
    ref_default = None
    if rank == 0 and not args.no_reference_default and args.noise_iters > 0:
        cfg_r = C.default_runtime_config(steps=args.sampler_steps, batch_size=1, noise_iters=args.noise_iters, gpu=local_rank)
        one = [parallel.slice_batch(batches[i % len(batches)], i % max(G, 1), i % max(G, 1) + 1) for i in range(5)]
        with contextlib.redirect_stdout(sys.stderr):                       # get_init_noise prints the scores like the reference
            for b in one[:2]:
                pipeline.predict(cfg_r, model, sampler, dict(b), dev)      # warm: captures the B = 1 graphs
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for b in one[2:]:
                pipeline.predict(cfg_r, model, sampler, dict(b), dev)
            torch.cuda.synchronize()
            dt_ref = (time.perf_counter() - t1) / len(one[2:])
            # the same workload through the in-flight driver (three single-image batches on three launch streams): what a
            # maintainer who hands predict_many the dataloader's images three at a time gets — reported beside, never as, the
            # one-image-at-a-time figure
            six = [parallel.slice_batch(batches[i % len(batches)], i % max(G, 1), i % max(G, 1) + 1) for i in range(6)]
            pipeline.predict_many(cfg_r, model, sampler, [dict(b) for b in six[:3]], dev, in_flight=3, fuse=1)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            pipeline.predict_many(cfg_r, model, sampler, [dict(b) for b in six], dev, in_flight=3, fuse=1)
            torch.cuda.synchronize()
            dt_ref3 = (time.perf_counter() - t2) / len(six)
        unet_g, enc_g, dec_g = WORK.get(args.size, WORK[512])
        calls = args.sampler_steps + 2 * args.noise_iters
        fpi_ref = (calls * unet_g + enc_g + dec_g + 7.2) * 1e9
        ref_default = {"images_per_s": 1.0 / dt_ref, "s_per_image": dt_ref, "images_per_s_three_in_flight": 1.0 / dt_ref3,
                       "unet_calls_per_image": calls,
                       "tflop_per_image": fpi_ref / 1e12, "frac_of_peak": fpi_ref / dt_ref / PEAK_BF16,
                       "workload": f"{args.size}x{args.size}, batch_size 1, noise_iters {args.noise_iters} (2 Euler steps + local "
                                   f"attention loss per candidate), {args.sampler_steps} steps, one image at a time "
                                   "(reference configs/test.yaml defaults)"}

    if rank == 0:
        images = args.steps * G
        value = images / elapsed
        fpi = flop_per_image(args.size, args.sampler_steps)
        traffic, traffic_file = measured_traffic()

        def cls(name, flops, nbytes, ms, launches):
            tf = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            return {"kernel": name, "bound": "mfma", "achieved": tf, "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                    "frac": tf / (PEAK_BF16 / 1e12), "launches": launches, "avg_launch_us": ms * 1e3 / max(launches, 1),
                    "algorithmic_gflop_per_launch": flops / max(launches, 1) / 1e9,
                    "algorithmic_bytes_per_launch": nbytes / max(launches, 1),
                    "algorithmic_hbm_gbps": nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                    "share_of_mfma_class_time": ms / max(conv_ms + gemm_ms + attn_ms, 1e-9)}

        import sgm.modules.hipnn as Hn
        conv = cls("3x3 convolution: wd::wconv3_kernel (256 px x 160 ch per workgroup, one per CU) + lg::lconv3_kernel (lean co-resident), both "
                   "LDS-staged patches" + (" / c3p::conv3p_kernel (GroupNorm+SiLU on the staged patch)" if Hn.FUSE_GN else "") +
                   " + g8::gemm8_kernel<CONV> (stride-2 gathers), UNet + VAE",
                   conv_flops, conv_bytes, conv_ms, conv_launches)
        traffic_if, traffic_if_file = measured_traffic_inflight_plans()
        conv.update({"traffic": traffic, "traffic_source": traffic_file,
                     "traffic_inflight_plans": traffic_if, "traffic_inflight_plans_source": traffic_if_file,
                     "traffic_hbm_gbps": (traffic / (conv["avg_launch_us"] * 1e-6) / 1e9) if traffic else None,
                     "traffic_frac_of_hbm_peak": (traffic / (conv["avg_launch_us"] * 1e-6) / PEAK_HBM) if traffic else None,
                     "traffic_scope": "patch-staged 3x3 launches (wd::wconv3_kernel + lg::lconv3_kernel: %d of the %d launches of the class): algorithmic "
                                      "%.1f MB per launch" % (int(c3p_launches), conv_launches, c3p_bytes / max(c3p_launches, 1) / 1e6),
                     "measured_on": "one eager single-stream pass of one local batch right after the timed region: HIP "
                                    "events around every launch, each kernel alone on the chip (the timed region replays "
                                    "hipGraphs with batches in flight, where launches of two streams overlap)",
                     "whole_path_frac_of_peak": value * fpi / (world * PEAK_BF16),
                     "whole_path_frac_of_fp8_peak": (value * fpi / (world * 2 * PEAK_BF16)) if args.fp8 else None})
        cfg_id = ("BASELINE.json configs[1]" if (args.size, args.batch, args.chars, args.sampler_steps, G, args.fp8) == (512, 4, 9, 50, 4 * world, False)
                  else "BASELINE.json configs[4]" if (args.fp8 and (args.size, args.sampler_steps, G, world) == (512, 50, 32, 8))
                  else ("BASELINE.json configs[4] arithmetic (fp8 MFMA attention / linear path), per-GPU share: batches of 4 of the 32 images "
                        "that config shards over 8 GPUs, on %d GPU(s)" % world) if (args.fp8 and (args.size, args.batch, args.sampler_steps) == (512, 4, 50))
                  else "BASELINE.json configs[2]" if (args.size, args.sampler_steps, G, world) == (512, 50, 64, 8)
                  else "BASELINE.json configs[3]" if (args.size, args.batch, args.chars, args.sampler_steps) == (768, 8, 12, 50)
                  else "non-baseline shape")
        per_rank = G // world
        line = {
            "metric": f"{args.size}x{args.size} {args.sampler_steps}-step denoised images/sec (UDiffText hot path: conditioner + "
                      f"{args.sampler_steps} CFG Euler steps + VAE decode)",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak" if weak else "strong",
            "vs_baseline": None, "data": "synthetic",
            "dtype": "fp8" if args.fp8 else "bf16",
            "dtype_note": ("MX8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4: e4m3 elements, one E8M0 scale per 32 K-elements of an activation "
                           "row written by the PRODUCING kernel's epilogue, per-output-channel weight scales) for every linear of the "
                           "640- / 1280-channel transformer blocks (q|k|v, to_out, GEGLU, ff.net[2], proj_out: 69 % of the linear "
                           "FLOPs; the 320-channel level's K = 320 projections are epilogue-bound and stay bf16, as do the "
                           "convolutions) and for the self-attention's Q K^T and P V at all three levels (q, k block-scaled along the "
                           "head dimension, v fixed-scale, probabilities as e4m3; text cross-attention bf16); fp32 accumulation, "
                           "statistics, softmax maxima / normalisation and sampler state") if args.fp8 else
                          "bf16 storage + MFMA, fp32 accumulation / statistics / softmax / sampler state",
            "value_one_batch": (value if (args.in_flight == 1 and args.fuse == 1) else other_modes.get("one_batch_at_a_time")),
            "value_one_batch_note": "the same K steps with ONE batch of --batch images on the GPU at a time (no batches in "
                                    "flight): the number that does not depend on how launch streams share the chip",
            "images_per_s_reference_default": ref_default,
            "images_per_s_by_launch_mode": dict(other_modes, **{"value": value}),
            "unet_ms_per_sampler_step": unet_ms,
            "unet_ms_note": f"one batch of {args.batch} alone on the whole GPU (latency of one UNet call on its CFG pair, one launch "
                            "stream, + the fused CFG/Euler update); with batches in flight the per-batch cost is lower",
            "config": {"workload": f"{args.size}x{args.size}, {args.sampler_steps} Euler/DDIM(eta 0) steps, CFG 5.0, {args.chars}-char "
                                   f"labels, noise_iters 0; global batch {G} per step sharded over {world} GPU(s) = {per_rank} "
                                   f"images per GPU in micro-batches of {args.batch}: every UNet call runs ONE batch of "
                                   f"{args.batch} ({2 * args.batch * max(args.fuse, 1)} samples with CFG); {args.in_flight} such "
                                   f"batch(es) in flight per GPU on separate launch streams; " + cfg_id,
                       "global_batch": G, "parallelism": f"dp{world} (images sharded contiguously, per-image seeds, one "
                                                         "all-gather of frames per global batch)",
                       "weights": "synthetic (name-keyed recipe), 1361.2 M parameters",
                       "launch": "hipGraph replay of the sampler steps" if graphs_on else "eager kernel launches",
                       "in_flight": args.in_flight, "batches_per_unet_call": max(args.fuse, 1) if args.fuse else "auto"},
            "roofline": conv,
            "roofline_classes": {
                "gemm": cls("linears + 1x1 convolutions: lg::lgemm_kernel (lean co-resident family; plain / GEGLU / LayerNorm-folded) "
                            "+ g8::gemm8_kernel (two-source 1x1, transposed, fp32 outputs)", gemm_flops,
                            gemm_bytes, gemm_ms, gemm_launches),
                "attention": cls("flash attention: " + ("attn_d64_mx8_kernel (UNet self-attention on e4m3 operands, head_dim 64; priced against the bf16 peak like the class)"
                                                 if args.fp8 and Hnn_fp8_attention() else "attn_d64_v2_kernel (UNet self-attention, head_dim 64)")
                                 + " + attn_d512_q64_kernel (VAE mid block, one head of 512)", attn_flops, attn_bytes, attn_ms, attn_launches)},
        }
        if world == 1 and not args.no_extra_configs and cfg_id == "BASELINE.json configs[1]":
            # the other single-GPU configurations of BASELINE.json on the same box, back to back with the line above (the GPU is idle
            # here: this process only holds memory); `value` stays config #2
            line["config5"] = extra_config(["--fp8"], args.steps, args.warmup, value)
            line["config4"] = extra_config(["--size", "768", "--batch", "8", "--chars", "12"], 3, 1)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.size, args.chars, args.sampler_steps, config1=args.cpu_config1)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
