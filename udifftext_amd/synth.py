"""Deterministic synthetic weights and inputs (there are no checkpoints or datasets in this environment).

``synthetic_tensor(name, shape)`` is a pure function of the parameter NAME and SHAPE (counter-based
Philox stream keyed by crc32(name), uniform samples), so the reference model (golden generation), the
CPU oracle and the HIP engine can all be filled with bit-identical fp32 weights without shipping them.

Scaling keeps activations O(1) through the network: matrices / conv kernels are fan-in scaled, norm
scales are 1 +- 0.1, biases are small; the layers the reference zero-initialises
(openaimodel.py:222-230,539; attention.py:131-136,391-395) get NON-zero weights — otherwise every output is
trivially zero.

``synthetic_batch`` reproduces the batch-dict contract of dataset/dataloader.py:274-287 / demo.py:87-98.
"""
from __future__ import annotations

import string
import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch

CHARSET = string.printable[:-6]   # encoders/modules.py:1097


def _uniform(name: str, n: int) -> np.ndarray:
    rng = np.random.Generator(np.random.Philox(key=zlib.crc32(name.encode("utf-8"))))
    return rng.random(n, dtype=np.float32) * 2.0 - 1.0   # U(-1, 1)


def synthetic_tensor(name: str, shape: Tuple[int, ...]) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u = _uniform(name, n)
    if len(shape) <= 1:
        if name.endswith("weight"):          # every 1-D ".weight" in this model is a norm scale
            v = 1.0 + 0.1 * u
        else:                                # biases
            v = 0.05 * u
    elif name.endswith("pos_embed") or name.endswith("pos_queries"):      # PARSeq's learned positions (OCR scorer)
        v = 0.5 * u
    elif "embedding.weight" in name:         # nn.Embedding
        v = u * np.float32(np.sqrt(3.0))
    else:
        fan_in = int(np.prod(shape[1:]))
        v = u * np.float32(np.sqrt(3.0 / fan_in))
    return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).reshape(shape)


# buffers that are computed by constructors, never random
_COMPUTED = ("denoiser.sigmas", "loss_fn.g_kernel", "pos_embedding.pe")


def is_computed_buffer(name: str) -> bool:
    return any(name.endswith(s) for s in _COMPUTED)


def fill_module_(module: torch.nn.Module, prefix: str = "") -> None:
    """In-place fill of every parameter of ``module`` (names as in its state_dict, with ``prefix``)."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            full = prefix + name
            if is_computed_buffer(full):
                continue
            p.copy_(synthetic_tensor(full, tuple(p.shape)).to(p.device, p.dtype))


def synthetic_state_dict(shapes: Iterable[Tuple[str, Tuple[int, ...]]]) -> Dict[str, torch.Tensor]:
    return {n: synthetic_tensor(n, tuple(s)) for n, s in shapes if not is_computed_buffer(n)}


# ------------------------------------------------------------------------------------------------ inputs
LABELS = {4: "TEXT", 9: "Diffusion", 12: "MI355XNative"}


def synthetic_label(n_chars: int, index: int = 0) -> str:
    base = LABELS.get(n_chars)
    if base is None:
        base = "".join(CHARSET[(7 * i + 3) % len(CHARSET)] for i in range(n_chars))
    if index == 0:
        return base
    # rotate so that different images of a batch render different strings of the same length
    k = index % len(base)
    return base[k:] + base[:k]


def synthetic_batch(batch_size: int, height: int, width: int, n_chars: int, seed: int = 0, max_len: int = 12) -> dict:
    """CPU fp32 batch dict: image in [-1,1], one centred box mask (25% x 75%), masked = image*(1-mask)."""
    g = torch.Generator().manual_seed(seed)
    image = torch.rand((batch_size, 3, height, width), generator=g) * 2.0 - 1.0
    mask = torch.zeros((batch_size, 1, height, width))
    bh, bw = height // 4, (width * 3) // 4
    top, left = (height - bh) // 2, (width - bw) // 2
    mask[:, :, top:top + bh, left:left + bw] = 1.0
    labels = [synthetic_label(n_chars, i) for i in range(batch_size)]
    seg_mask = torch.zeros((batch_size, max_len))
    seg_mask[:, :n_chars] = 1.0
    return {
        "image": image,
        "mask": mask,
        "masked": image * (1.0 - mask),
        "seg_mask": seg_mask,
        "label": labels,
        "txt": [f'"{s}"' for s in labels],
        "target_size_as_tuple": torch.tensor([[height, width]] * batch_size),
        "name": [str(i) for i in range(batch_size)],
        "r_bbox": torch.tensor([[top, top + bh, left, left + bw]] * batch_size),
    }
