"""Inputs of the attend-and-excite goldens (G13 / G13s, tests/golden/aae_golden.npz) — shared by the generator
(tests/golden/make_golden.py --g13) and the tests.  Data only."""
import torch

from udifftext_amd import synth


def aae_batch() -> dict:
    """synthetic_batch(1, 128, 128, 4, seed=13) with masks that DECIDE the hard selections of get_min_local_loss.  With the synthetic
    weights every t_attn map is nearly uniform over the 12 tokens (largest probability 0.0837 against 1 / 12 = 0.0833): the margins of
    the loss's min over tokens / max over pixels are 1e-6 .. 1e-4 of the values — below what fp32 reproduces across CPUs (the same
    oracle on another host picked another token; the gradient of the hard loss is then a different function).  So ONE token is scored
    (seg_mask = e_0) and the mask covers one 8 x 8 block of the image = one cell of the 16 x 16 maps: the selections are decided by
    the inputs, everything downstream (blur stencil, head mean, the whole reverse pass) is what the golden pins.  Arg-max / arg-min
    themselves are tested on given probabilities (tests/test_backward_gpu.py::test_local_loss_backward_vs_autograd)."""
    batch = synth.synthetic_batch(1, 128, 128, 4, seed=13)
    mask = torch.zeros_like(batch["mask"])
    mask[:, :, 56:64, 40:48] = 1.0                       # latent cell (7, 5)
    batch["mask"] = mask
    batch["masked"] = batch["image"] * (1.0 - mask)
    seg = torch.zeros_like(batch["seg_mask"])
    seg[:, 0] = 1.0
    batch["seg_mask"] = seg
    return batch


def aae_functional_weights(shape, k: int) -> torch.Tensor:
    """fixed pseudo-random weights R_k of the smooth functional sum_k <R_k, attn_map_k> / count (G13s)"""
    return torch.randn(tuple(shape), generator=torch.Generator().manual_seed(1300 + k))


def train_batch() -> dict:
    """the batch of the training-step goldens (G14 / G14s, tests/golden/train_golden.npz): two 128 x 128 images, 4 characters, and
    per-character segment maps ``seg`` [B, 12, 128, 128]: the mask box cut into four equal cells, one per character"""
    batch = synth.synthetic_batch(2, 128, 128, 4, seed=14)
    B = 2
    seg = torch.zeros((B, 12, 128, 128))
    top, bottom, left, right = [int(v) for v in batch["r_bbox"][0]]
    cw = (right - left) // 4
    for l in range(4):
        seg[:, l, top:bottom, left + l * cw:left + (l + 1) * cw] = 1.0
    batch["seg"] = seg
    return batch


def sub(t: torch.Tensor, n: int = 512):
    """strided sub-sample of a flattened tensor (<= n values) — how the goldens store large gradients"""
    f = t.detach().float().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n]
