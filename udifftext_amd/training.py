"""One training step of the text cross-attention on the HIP path (second half).

Reference: ``DiffusionEngine.forward / shared_step / training_step / configure_optimizers`` (sgm/models/diffusion.py:138-172,202-222)
with ``FullLoss.__call__`` (sgm/modules/diffusionmodules/loss.py:131-176): noise the latent at a sampled sigma, run the denoiser,

    loss = mean_b mean(w (D(x + n sigma) - x)^2)  +  lambda_local_loss * mean_b get_local_loss(t_attn maps, seg, seg_mask),

back-propagate to the parameters whose names contain an ``opt_keys`` entry (configs/train/textdesign_sd_2.yaml:4-6: ``t_attn``,
``t_norm`` — 75.9 M of the UNet's 866 M), average the gradients over the data-parallel ranks, AdamW step, lr = base * 0.95^epoch.

Here: the tape-mode forward and the written-out reverse pass of ``udifftext_amd.backward`` (dX through every layer, dW only where
the reference trains), the two loss seeds and the optimiser update as HIP kernels (csrc/backward.hip), one flat fp32 bucket per step
through ``torch.distributed`` (RCCL on the GPUs: reduce-scatter + all-gather, the bandwidth-optimal form on the xGMI mesh; gloo's
all_reduce in the CPU tests).  The OCR / style losses (ocr_enabled / style_enabled: False in every shipped config), EMA and the
Lightning loop stay out of scope.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import backward, ops, rng


# ------------------------------------------------------------------------------------------------ the loss and its gradients
def training_loss_and_grads(engine, z: torch.Tensor, cond: dict, seg: torch.Tensor, seg_mask: torch.Tensor,
                            sigma_idx: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
                            want_grads: bool = True):
    """FullLoss.__call__ on latents z fp32 [B, 4, h, w] with conditioning ``cond`` ({"concat": [B, 5, h, w], "t_crossattn":
    [B, L, Dc]}), character segment maps seg fp32 [B, seg_l, Hs, Ws] and seg_mask [B, seg_l].
    sigma_idx int64 [B] (default: DiscreteSampling's torch.randint draw) and noise [B, 4, h, w] (default: one CPU-generator randn,
    rng.randn_on) are the step's random draws.  Returns (loss_dict, grads): loss_dict as the reference's (``loss/diff_loss``,
    ``loss/local_loss``, ``loss/full_loss``: 0-dim fp32 tensors), grads = {state-dict name: fp32 gradient of loss/full_loss} for the
    t_attn / t_norm parameters (None when want_grads is False)."""
    loss_fn = engine.loss_fn
    B = z.shape[0]
    dev = z.device
    tape, noised, sigma = training_tape(engine, z, cond, sigma_idx, noise)
    z = z.float().contiguous()
    loss_diff, d_eps = ops.diff_loss_grad(tape.eps, noised, z, sigma)
    used = [it for it in tape.maps if it["size"] >= loss_fn.min_attn_size]
    lam = float(loss_fn.lambda_local_loss)
    loss_local = torch.zeros((B,), dtype=torch.float32, device=dev)
    if used:
        segf, segm = seg.float().contiguous(), seg_mask.float().contiguous()
        gk = loss_fn.g_kernel[0, 0].reshape(9).float().contiguous()
        for it in used:
            it["d_probs"] = torch.zeros_like(it["attn_map"])
            ops.local_loss_seg_bwd(it["attn_map"], segf, segm, gk, it["d_probs"], loss_local, it["heads"], it["size"],
                                   lam / (len(used) * B))
    diff = loss_diff.mean()
    local = loss_local.mean() / max(len(used), 1)
    loss_dict = {"loss/diff_loss": diff, "loss/local_loss": local, "loss/full_loss": diff + lam * local}
    if not want_grads:
        return loss_dict, None
    grads: Dict[str, torch.Tensor] = {}
    tape.backward(d_eps, param_grads=grads)
    return loss_dict, grads


def training_tape(engine, z: torch.Tensor, cond: dict, sigma_idx: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None):
    """the training forward: noise z at the sampled sigmas, DiscreteDenoiser's input scaling, tape-mode UNet with its output head.
    Returns (tape, noised fp32 [B, 4, h, w], sigma fp32 [B])"""
    from sgm.modules.diffusionmodules.openaimodel import CPAD
    dev = z.device
    B, _, h, w = z.shape
    den = engine.denoiser
    unet = engine.model.diffusion_model
    table = den.sigmas.to(dev).float()
    if sigma_idx is None:
        sigma_idx = torch.randint(0, table.numel(), (B,))               # DiscreteSampling.__call__ (CPU draw, reference order)
    sigma_idx = sigma_idx.to(dev).long()
    sigma = table[sigma_idx].contiguous()                                # (the sampled sigmas lie on the denoiser's table)
    if noise is None:
        noise = rng.randn_on((B, 4, h, w), dev)
    z = z.float().contiguous()
    noised = z.clone()
    sig_host = [float(s) for s in sigma.cpu()]
    for b in range(B):                                                   # noised = z + n sigma_b
        ops.axpy_(noised[b], noise[b].float().contiguous(), sig_host[b])
    scaled = noised.clone()
    for b in range(B):                                                   # network input = noised * c_in(sigma_b) (DiscreteDenoiser)
        ops.axpy_(scaled[b], scaled[b], 1.0 / (sig_host[b] ** 2 + 1.0) ** 0.5 - 1.0)
    xin = ops.nchw_to_nhwc(torch.cat((scaled, cond["concat"].float()), dim=1).contiguous(), CPAD)
    tape = backward.UNetTape(unet, xin, sigma_idx.float(), cond["t_crossattn"], with_head=True)
    return tape, noised, sigma


def trainable_parameters(engine, opt_keys: Optional[List[str]] = None) -> List[Tuple[str, torch.nn.Parameter]]:
    """the (state-dict name, parameter) pairs DiffusionEngine.configure_optimizers selects (diffusion.py:204-217): names under
    ``model.`` that contain an opt_keys entry, in module order"""
    keys = opt_keys if opt_keys is not None else (engine.opt_keys or [])
    return [("model." + n, p) for n, p in engine.model.named_parameters() if any(k in n for k in keys)]


# ------------------------------------------------------------------------------------------------ data-parallel gradient average
def allreduce_gradients(grads: Dict[str, torch.Tensor], names: List[str], dist=None, force: bool = False) -> None:
    """average the gradients over the ranks, in place: ONE flat fp32 bucket in the order ``names`` (the same on every rank).  On RCCL:
    reduce-scatter + all-gather of the padded bucket — on the xGMI full mesh both are direct peer exchanges, 2 (N - 1) / N of the
    bucket per GPU; with gloo (CPU tests): all_reduce."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return                                                           # (force: run the collectives in a world of one — tests)
    world = dist.get_world_size()
    flat = torch.cat([grads[n].reshape(-1) for n in names])
    n = flat.numel()
    if dist.get_backend() == "nccl":
        pad = (-n) % world
        if pad:
            flat = torch.cat([flat, flat.new_zeros(pad)])
        shard = torch.empty((flat.numel() // world,), dtype=flat.dtype, device=flat.device)
        dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM)
        dist.all_gather_into_tensor(flat, shard)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    o = 0
    for nm in names:
        g = grads[nm]
        g.copy_(flat[o:o + g.numel()].reshape(g.shape))
        o += g.numel()
        if g.is_cuda:
            ops.axpy_(g.reshape(-1), g.reshape(-1), 1.0 / world - 1.0)     # g *= 1 / world
        else:
            g.mul_(1.0 / world)                                          # (CPU tests: no HIP kernels)


# ------------------------------------------------------------------------------------------------ optimiser
class AdamW:
    """torch.optim.AdamW (the reference's default optimiser, diffusion.py:49-51) over named fp32 parameters, stepped by udt_adamw_f32;
    ``set_epoch`` applies configure_optimizers' LambdaLR (lr = base * 0.95^epoch, diffusion.py:220)"""

    def __init__(self, named_params: List[Tuple[str, torch.nn.Parameter]], lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2):
        self.named = list(named_params)
        self.base_lr, self.lr, self.betas, self.eps, self.weight_decay = lr, lr, betas, eps, weight_decay
        self.step_count = 0
        self.state = {n: (torch.zeros_like(p.data, dtype=torch.float32), torch.zeros_like(p.data, dtype=torch.float32)) for n, p in self.named}

    def set_epoch(self, epoch: int) -> None:
        self.lr = self.base_lr * 0.95 ** epoch

    def step(self, grads: Dict[str, torch.Tensor], grad_scale: float = 1.0) -> None:
        self.step_count += 1
        for n, p in self.named:
            m, v = self.state[n]
            ops.adamw_(p, grads[n].contiguous(), m, v, self.step_count, self.lr, self.betas, self.eps, self.weight_decay, grad_scale)


def training_step(engine, optimizer: AdamW, z: torch.Tensor, cond: dict, seg: torch.Tensor, seg_mask: torch.Tensor, dist=None,
                  sigma_idx: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None) -> dict:
    """loss + gradients + rank average + AdamW update of the t_attn / t_norm parameters; returns the loss dict"""
    loss_dict, grads = training_loss_and_grads(engine, z, cond, seg, seg_mask, sigma_idx=sigma_idx, noise=noise)
    names = [n for n, _ in optimizer.named]
    allreduce_gradients(grads, names, dist)
    optimizer.step(grads)
    return loss_dict
