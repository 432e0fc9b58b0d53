"""Inference driver: the reference's ``util.py`` (init_model / init_sampling / prepare_batch, :7-78) and
``test.py:predict`` (:19-40) restated against the MI355X ``sgm`` package, for synthetic batches.

The reference's own ``test.py`` needs datasets, checkpoints and packages that do not exist here
(SURVEY.md §8b); this module is the shipped equivalent of its hot path and is what bench.py times.
"""
from __future__ import annotations

import contextlib
import time
from typing import Optional, Tuple

import torch

import udifftext_amd  # noqa: F401  (puts the sgm mirror on sys.path)
from udifftext_amd import config as C
from udifftext_amd import rng, synth


def build_engine(device: torch.device, synthetic_weights: bool = True, verbose: bool = False):
    """instantiate the DiffusionEngine from the (code-built) model config, fill deterministic synthetic weights
    (there are no checkpoints here), move it to the GPU, eval + freeze — reference util.py:7-22."""
    from sgm.util import instantiate_from_config, skip_param_init
    t0 = time.time()
    cfg = C.default_model_config()
    if synthetic_weights:
        with skip_param_init():
            model = instantiate_from_config(cfg.model)
        synth.fill_module_(model)
    else:
        model = instantiate_from_config(cfg.model)
    model.to(device)
    model.eval()
    model.freeze()
    if verbose:
        n = sum(p.numel() for p in model.parameters())
        print(f"[pipeline] engine with {n / 1e6:.1f} M parameters ready in {time.time() - t0:.1f}s")
    return model


def init_sampling(steps: int, scale: float, device: torch.device, verbose: bool = False):
    """reference util.py:24-47: EulerEDMSampler + LegacyDDPMDiscretization + VanillaCFG(scale)"""
    from sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    return EulerEDMSampler(
        num_steps=steps,
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"},
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": scale}},
        s_churn=0.0, s_tmin=0.0, s_tmax=999.0, s_noise=1.0, verbose=verbose, device=device)


def _copy_batch(batch: dict) -> dict:
    out = {}
    for k, v in batch.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.clone()
        elif isinstance(v, (tuple, list)):
            out[k] = list(v)
        else:
            out[k] = v
    return out


def prepare_batch(batch: dict, device: torch.device) -> Tuple[dict, dict]:
    """reference util.py:62-78: tensors to the device; the unconditional batch has empty labels / prompts"""
    for k, v in batch.items():
        if isinstance(v, torch.Tensor):
            batch[k] = v.to(device)
    buc = _copy_batch(batch)
    buc["txt"] = batch["ntxt"] if "ntxt" in batch else ["" for _ in batch["txt"]]
    if "label" in batch:
        buc["label"] = ["" for _ in batch["label"]]
    # host-side knowledge for the conditioner: every tensor of buc is a clone of batch's (it then shares the masked-image
    # encoder pass between c and uc without comparing the two tensors on the device)
    # The marker is only as good as the tensors it was made for: it records (identity, version counter) of both sides per key, and
    # the conditioner falls back to comparing the tensors when either was replaced or written to since (e.g. a caller that gives the
    # unconditional batch a different masked image between prepare_batch and get_unconditional_conditioning).
    buc["_udt_clone_of"] = batch
    buc["_udt_changed"] = ("txt", "label")
    buc["_udt_clone_state"] = {k: (id(buc[k]), buc[k]._version, id(v), v._version) for k, v in batch.items() if isinstance(v, torch.Tensor)}
    return batch, buc


@torch.no_grad()
def predict(cfgs, model, sampler, batch: dict, device: Optional[torch.device] = None):
    """reference test.py:19-40 -> (samples in [0,1] fp32 NCHW, latent z)"""
    device = device or next(model.parameters()).device
    batch, batch_uc = prepare_batch(batch, device)
    c, uc = model.conditioner.get_unconditional_conditioning(
        batch, batch_uc=batch_uc, force_uc_zero_embeddings=cfgs.force_uc_zero_embeddings)
    x = sampler.get_init_noise(cfgs, model, cond=c, batch=batch, uc=uc)
    z = sampler(model, x, cond=c, batch=batch, uc=uc, init_step=cfgs.init_step, aae_enabled=cfgs.aae_enabled,
                detailed=cfgs.detailed)
    img = model.decode_first_stage(z)
    return torch.clamp((img + 1.0) / 2.0, min=0.0, max=1.0), z


IN_FLIGHT = 3      # launch streams sampling concurrently in predict_many (same-box on MI355X, batches of 4 at 512x512:
#                    1 -> 5.7, 2 -> 7.13, 3 -> 7.67, 4 -> 6.0 images/s; 1 = one at a time)
FUSE = 0           # batches concatenated into one sampling batch per stream; 0 = automatic: the list is spread over
#                    the streams, at most 4 batches per sampling batch (dynamic batching: 8 images per UNet call
#                    cost 2.87 ms per step and image against 3.6 ms for 4, MI355X; per-sample statistics only, so every
#                    image's result depends on its own inputs alone)


def _cat_cond(conds):
    out = {}
    for k in conds[0]:
        v = conds[0][k]
        out[k] = torch.cat([c[k] for c in conds], 0) if isinstance(v, torch.Tensor) else v
        if isinstance(v, torch.Tensor) and all(getattr(c[k], "_udt_all_zero", False) for c in conds):
            out[k]._udt_all_zero = True          # (host-side tag of force-zeroed embeddings, GeneralConditioner.forward)
    return out


_LANE_STREAMS: dict = {}


def _lane_streams(device: torch.device, n: int):
    key = (device.index, n)
    if key not in _LANE_STREAMS:
        _LANE_STREAMS[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
    return _LANE_STREAMS[key]


def predict_many(cfgs, model, sampler, batches, device: Optional[torch.device] = None, in_flight: Optional[int] = None,
                 fuse: Optional[int] = None, image_seeds: Optional[list] = None):
    """``predict`` over a list of batches in throughput mode: ``fuse`` consecutive batches are concatenated into one
    sampling batch, and up to ``in_flight`` such batches are processed concurrently, each on its own launch stream
    (a lane): conditioning (VAE encoder, label encoder), the sampling loop's hipGraph replays (EulerEDMSampler.sample_lane)
    and the VAE decode of a batch are enqueued back to back on its lane's stream; the lanes run free of each other and the
    host only synchronises once, at the end.
    Conditioning and noise draws stay per input batch, in the order and from the CPU generator that calling
    ``predict`` batch by batch would use; decoding runs on the fused batch.
    ``image_seeds[k]`` (optional): one seed per image of batch k — its draws then come from per-image generators
    (``rng.per_image``), independent of batching and sharding.  Returns [(samples, z), ...] in input order."""
    from udifftext_amd import ops
    device = device or next(model.parameters()).device
    n = max(1, int(in_flight if in_flight is not None else IN_FLIGHT))
    f = int(fuse if fuse is not None else FUSE)
    if f <= 0:
        f = min(4, max(1, -(-len(batches) // n)))
    if cfgs.aae_enabled or cfgs.detailed:
        raise NotImplementedError("attend-and-excite / detailed dumps are out of scope (see EulerEDMSampler.__call__)")
    # (host logic only below; with a stub engine on the CPU — tests/test_parallel_cpu.py — there are no streams)
    gpu = torch.device(device).type == "cuda"
    main = torch.cuda.current_stream(device) if gpu else None
    lanes = (_lane_streams(device, n) if n > 1 else [main]) if gpu else [None] * n
    on = (lambda lane: torch.cuda.stream(lane)) if gpu else (lambda lane: contextlib.nullcontext())
    after = (lambda a, b: a.wait_stream(b)) if gpu else (lambda a, b: None)       # stream a continues after stream b
    out, checks, keep = [], [], []
    # FREE-RUNNING lanes (round 4): sampling batch u runs on lane u % n — conditioning, the sampling loop's graph replays and
    # the decode are enqueued back to back on the lane's stream, with no event shared between lanes: a lane starts its next
    # batch the moment its previous decode is done, and the conditioning / decode phases of one lane (eager launches, ~25 ms)
    # run beside the other lanes' sampling.  Nothing below synchronises the host: the noise draws travel through pinned
    # memory (rng.randn_on), the zero-context test and the c / uc image comparison are host-side flags.  (Round 3 ran the
    # lanes in lock-stepped groups with a cross-lane join before and after every sampling loop: ~35 ms of each 1.28 s group
    # had no sampling running, and a slow lane held the others.)
    lane_sampler = getattr(sampler, "sample_lane", None)
    units = [list(range(i, min(i + f, len(batches)))) for i in range(0, len(batches), f)]
    # lanes in use: as few as keep every lane equally loaded — 4 sampling batches on 3 lanes run as 2 + 2 on TWO lanes (each planned
    # for half of the CUs), not as 2 + 1 + 1 with the 4th batch alone on a lane planned for a third of the chip while two lanes idle
    rounds = -(-len(units) // n) if units else 1
    n_used = max(1, -(-len(units) // rounds)) if units else 1
    for lane in lanes[:n_used]:
        after(lane, main)
    for u, idx in enumerate(units):
        lane = lanes[u % n_used]
        xs, cs, ucs, sizes = [], [], [], []
        with on(lane), ops.launch_context(cu_share=n_used):
            for gi in idx:
                b, buc = prepare_batch(batches[gi], device)
                seeds = image_seeds[gi] if image_seeds is not None else None
                with (rng.per_image(seeds) if seeds is not None else contextlib.nullcontext()):
                    c, uc = model.conditioner.get_unconditional_conditioning(
                        b, batch_uc=buc, force_uc_zero_embeddings=cfgs.force_uc_zero_embeddings)
                    xs.append(sampler.get_init_noise(cfgs, model, cond=c, batch=b, uc=uc))
                cs.append(c)
                ucs.append(uc)
                sizes.append(xs[-1].shape[0])
            if any(x.shape[1:] != xs[0].shape[1:] for x in xs):
                raise ValueError("batches fused into one sampling batch need one image size (use fuse=1)")
            fx = torch.cat(xs, 0) if len(xs) > 1 else xs[0]
            fc = _cat_cond(cs) if len(cs) > 1 else cs[0]
            fuc = _cat_cond(ucs) if len(ucs) > 1 else ucs[0]
            if lane_sampler is not None:
                z = lane_sampler(model, fx, fc, fuc, slot=u % n_used, n_lanes=n_used, init_step=cfgs.init_step, deferred_checks=checks)
            else:       # (stub samplers of the CPU tests: the group interface, one batch at a time)
                z = sampler.sample_in_flight(model, [fx], [fc], [fuc], init_step=cfgs.init_step, deferred_checks=checks)[0]
            img = torch.clamp((model.decode_first_stage(z) + 1.0) / 2.0, min=0.0, max=1.0)
        keep.append((xs, cs, ucs, fx, fc, fuc, z))     # tensors of a lane stream: alive until the final synchronisation
        o = 0
        for nb in sizes:
            out.append((img[o:o + nb], z[o:o + nb]))
            o += nb
    for lane in lanes:
        after(main, lane)
    for chk in checks:
        chk()
    if gpu:
        ops.check_async_errors()
    if hasattr(sampler, "release_retired"):             # (the checks above synchronised every lane: evicted graph runners may go)
        sampler.release_retired()
    del keep
    return out
