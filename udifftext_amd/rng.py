"""The path's noise source.

The reference draws every inference-time random tensor with ``torch.randn`` on the CPU default generator and moves it
to the device (posterior noise of ``c``, posterior noise of ``uc``, then ``noise_iters + 1``
initial-noise candidates — distributions.py:38-40, sampling.py:269,311).  ``randn`` is that draw and the default.

For sharded sampling the draws must not depend on how images are spread over ranks or grouped into
batches: inside ``per_image(seeds)`` every image of the batch owns a CPU generator seeded with ITS seed, and each draw
of shape ``[B, ...]`` is the concatenation of one ``[1, ...]`` draw per image — so image i sees exactly the stream a
batch-1 run of the reference seeded with ``seeds[i]`` would see, whatever the world size.
"""
from __future__ import annotations

import contextlib
import threading
from typing import List, Optional, Sequence

import torch


class _State(threading.local):
    gens: Optional[List[torch.Generator]] = None


_state = _State()


def randn(shape: Sequence[int]) -> torch.Tensor:
    """CPU fp32 standard-normal tensor of ``shape`` from the active noise source"""
    shape = tuple(int(s) for s in shape)
    gens = _state.gens
    if gens is None:
        return torch.randn(shape)
    if len(gens) != shape[0]:
        raise ValueError(f"per-image noise source holds {len(gens)} generators, draw asks for batch {shape[0]}")
    return torch.cat([torch.randn((1,) + shape[1:], generator=g) for g in gens], 0)


def randn_on(shape: Sequence[int], device) -> torch.Tensor:
    """the same draw as ``randn`` (same generator(s), same values), delivered on ``device``: drawn into page-locked host
    memory and copied with a non-blocking, stream-ordered transfer.  A pageable ``.to(device)`` blocks the host until every
    kernel already queued on the current stream has run — with several batches in flight (pipeline.predict_many) that stalls
    the launch thread for a whole sampling loop.  The pinned block stays alive until the copy has run (PyTorch's caching
    host allocator records the stream use)."""
    device = torch.device(device)
    if device.type != "cuda":
        return randn(shape).to(device)
    shape = tuple(int(s) for s in shape)
    gens = _state.gens
    host = torch.empty(shape, dtype=torch.float32, pin_memory=True)
    if gens is None:
        torch.randn(shape, out=host)
    else:
        if len(gens) != shape[0]:
            raise ValueError(f"per-image noise source holds {len(gens)} generators, draw asks for batch {shape[0]}")
        for i, g in enumerate(gens):
            torch.randn((1,) + shape[1:], generator=g, out=host[i:i + 1])
    return host.to(device, non_blocking=True)


@contextlib.contextmanager
def per_image(seeds: Sequence[int]):
    """draws inside come from one generator per image (seeded here; each context starts fresh streams)"""
    prev = _state.gens
    _state.gens = [torch.Generator().manual_seed(int(s)) for s in seeds]
    try:
        yield
    finally:
        _state.gens = prev
