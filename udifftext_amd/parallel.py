"""Multi-GPU driver: a GLOBAL batch of images is sharded contiguously over the ranks (one process per GPU, full weight
replica, no collective inside the denoising loop) and the decoded frames are collected with ONE all-gather per global
batch over RCCL/xGMI (BASELINE.json configs[2]: batch 64 over 8 GPUs).

The reference has no multi-GPU inference (configs/test.yaml:24 ``gpu: 0``); this is the data-parallel layer the north
star adds.  Every image owns a seed derived from (global seed, global image index) and draws its noise from its own
CPU generator (``udifftext_amd.rng``), so an image's result does not depend on the world size, on which rank samples
it, or on how the rank groups its shard into batches.  Works with any torch.distributed backend ("nccl" == RCCL on
ROCm; "gloo" in the CPU tests).
"""
from __future__ import annotations

import copy
from typing import Callable, List, Optional, Sequence, Tuple

import torch


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous [begin, end) slice of n_items owned by ``rank`` (sizes differ by at most one)"""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def image_seed(global_seed: int, image_index: int) -> int:
    """per-image seed, independent of the world size, so results do not depend on how images are sharded"""
    return (global_seed * 1000003 + image_index * 7919) % (2 ** 31 - 1)


def gather_frames(frames: torch.Tensor, dist=None) -> torch.Tensor:
    """[B_local, 3, H, W] on every rank -> [B_local * world, 3, H, W] on every rank (rank-major order)."""
    if dist is None or not dist.is_initialized():
        return frames
    # (a world of ONE still goes through the collective: the RCCL path then runs on a single-GPU box too — the GPU tests and
    #  bench.py's UDT_BENCH_FORCE_DIST use that; it is a device-to-device copy)
    world = dist.get_world_size()
    frames = frames.contiguous()
    out = torch.empty((world * frames.shape[0],) + tuple(frames.shape[1:]), dtype=frames.dtype, device=frames.device)
    dist.all_gather_into_tensor(out, frames)
    return out


def gather_ragged(frames: torch.Tensor, counts: List[int], dist) -> torch.Tensor:
    """all-gather when ranks hold different numbers of images (``counts[r]`` images on rank r): ONE collective on
    frames padded to the largest shard"""
    world = dist.get_world_size()
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(frames.shape[1:]), dtype=frames.dtype, device=frames.device)
    pad[:frames.shape[0]] = frames
    full = gather_frames(pad, dist).reshape((world, mx) + tuple(frames.shape[1:]))
    return torch.cat([full[r, :counts[r]] for r in range(world)], 0)


def slice_batch(batch: dict, begin: int, end: int) -> dict:
    """images [begin, end) of a batch dict (tensors and per-image lists are sliced, everything else is shared)"""
    n = len(batch["label"]) if "label" in batch else batch["image"].shape[0]
    out = {}
    for k, v in batch.items():
        if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == n:
            out[k] = v[begin:end]
        elif isinstance(v, (list, tuple)) and len(v) == n:
            out[k] = list(v[begin:end])
        else:
            out[k] = v
    return out


def predict_sharded(cfgs, model, sampler, global_batches: Sequence[dict], global_seeds: Sequence[int], dist=None,
                    micro_batch: int = 4, in_flight: Optional[int] = None, fuse: int = 1, device=None,
                    predict_many: Optional[Callable] = None) -> List[torch.Tensor]:
    """Sample every global batch (a batch dict for ALL its images, CPU or device tensors) across the ranks of
    ``dist`` and return, on every rank, the frames ``[N, 3, H, W]`` of each global batch in global image order.

    Rank r owns images ``shard_range(N, r, world)`` of every global batch, cuts its shard into micro-batches of at most
    ``micro_batch`` images (config #3: 8 images per GPU = two batches of 4, sampled concurrently on two launch
    streams) and hands them to ``predict_many`` (default: pipeline.predict_many) with per-image seeds; then ONE
    all-gather per global batch collects the decoded frames.  ``cfgs.batch_size`` is set per micro-batch."""
    if predict_many is None:
        from udifftext_amd import pipeline
        predict_many = pipeline.predict_many
    on = dist is not None and dist.is_initialized()
    rank = dist.get_rank() if on else 0
    world = dist.get_world_size() if on else 1
    micro, seeds, owner = [], [], []          # micro-batches of all global batches, in order
    counts_all = []
    for gi, (gb, gseed) in enumerate(zip(global_batches, global_seeds)):
        n = len(gb["label"]) if "label" in gb else gb["image"].shape[0]
        counts_all.append([shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)])
        begin, end = shard_range(n, rank, world)
        for a in range(begin, end, micro_batch):
            b = min(a + micro_batch, end)
            micro.append(slice_batch(gb, a, b))
            seeds.append([image_seed(gseed, i) for i in range(a, b)])
            owner.append(gi)
    # micro-batches of one size share cfgs (batch_size is read by get_init_noise); run them grouped by size
    results: List[Optional[torch.Tensor]] = [None] * len(micro)
    sizes = sorted({len(s) for s in seeds}, reverse=True)
    for sz in sizes:
        idx = [i for i, s in enumerate(seeds) if len(s) == sz]
        cfg = type(cfgs)(cfgs) if isinstance(cfgs, dict) else copy.copy(cfgs)      # never mutate the caller's config
        cfg.batch_size = sz
        outs = predict_many(cfg, model, sampler, [micro[i] for i in idx], device, in_flight=in_flight, fuse=fuse,
                            image_seeds=[seeds[i] for i in idx])
        for i, (smp, _) in zip(idx, outs):
            results[i] = smp
    frames = []
    for gi in range(len(global_batches)):
        mine = [results[i] for i in range(len(micro)) if owner[i] == gi]
        counts = counts_all[gi]
        if mine:
            local = torch.cat(mine, 0)
        else:
            # more ranks than images: an empty shard still joins the collective.  Its placeholder is built from the
            # batch's own metadata (this rank may not have sampled ANY image, e.g. a global batch of 4 on 8 GPUs)
            gb = global_batches[gi]
            Hh, Ww = (int(v) for v in gb["target_size_as_tuple"][0]) if "target_size_as_tuple" in gb else gb["image"].shape[-2:]
            ref = next((r for r in results if r is not None), None)
            dt = ref.dtype if ref is not None else torch.float32
            dv = ref.device if ref is not None else (torch.device(device) if device is not None else gb["image"].device)
            local = torch.zeros((0, 3, Hh, Ww), dtype=dt, device=dv)
        if not on:
            frames.append(local)
        elif len(set(counts)) == 1:
            frames.append(gather_frames(local, dist))
        else:
            frames.append(gather_ragged(local, counts, dist))
    return frames
