"""Multi-GPU driver pieces: images shard across ranks (one process per GPU, no collective inside the denoising
loop), decoded frames are collected with ONE all-gather per batch over RCCL/xGMI.

The reference has no multi-GPU inference (configs/test.yaml:24 ``gpu: 0``); this is the data-parallel layer the
north star adds.  The functions work with any torch.distributed backend ("nccl" == RCCL on ROCm; "gloo" in the
CPU tests).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

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
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return frames
    world = dist.get_world_size()
    frames = frames.contiguous()
    out = torch.empty((world * frames.shape[0],) + tuple(frames.shape[1:]), dtype=frames.dtype, device=frames.device)
    dist.all_gather_into_tensor(out, frames)
    return out


def gather_ragged(frames: torch.Tensor, counts: List[int], dist) -> torch.Tensor:
    """all-gather when ranks hold different numbers of images (``counts[r]`` images on rank r)"""
    world = dist.get_world_size()
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(frames.shape[1:]), dtype=frames.dtype, device=frames.device)
    pad[:frames.shape[0]] = frames
    full = gather_frames(pad, dist).reshape((world, mx) + tuple(frames.shape[1:]))
    return torch.cat([full[r, :counts[r]] for r in range(world)], 0)
