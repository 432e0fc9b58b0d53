"""Load-time preparation of a ``DiffusionEngine`` (SURVEY.md §8b "Checkpoint contract": internal repacking happens
after load, in a prepare step; §8f rank 3: checkpoint ingestion & weight packing).

``prepare(engine)`` does, once, what the first forward would otherwise do lazily:

* **dedup of the twin VAE** — the reference engine holds two copies of the autoencoder, ``first_stage_model.*`` and
  ``conditioner.embedders.2.model.*``, loaded from the same file (configs/test/textdesign_sd_2.yaml:70,91;
  sgm/models/diffusion.py:87-105).  When every tensor of the two is equal the LatentEncoder is pointed at
  ``first_stage_model``: one set of masters, one set of packed weights (the state dict keeps both prefixes).
* **packing** — every module's device layout (bf16, K-contiguous, fused q|k / k|v / GEGLU blocks, the 22 ``emb_layers``
  as one matrix) is derived now; modules whose weights are consumed only through a parent's fused pack are skipped.
* **free_masters=True** — the fp32 checkpoint-layout parameters of the packed modules are released (5.4 GB of the
  engine's 8.1 GB on the device) and the caches are frozen.  The engine then serves inference only: its state dict no
  longer holds those weights, so reload the checkpoint into a fresh engine to change them.
"""
from __future__ import annotations

from typing import Dict

import torch


def _same_weights(a: torch.nn.Module, b: torch.nn.Module) -> bool:
    sa, sb = a.state_dict(), b.state_dict()
    if list(sa.keys()) != list(sb.keys()):
        return False
    return all(sa[k].shape == sb[k].shape and sa[k].device == sb[k].device and bool(torch.equal(sa[k], sb[k])) for k in sa)


def prepare(engine, free_masters: bool = False, dedup_vae: bool = True) -> Dict[str, float]:
    """returns a report: {"vae_deduplicated", "packed_modules", "packed_bytes", "freed_bytes"}"""
    from sgm.modules import hipnn as H
    from sgm.modules.encoders.modules import LatentEncoder
    report = {"vae_deduplicated": 0, "packed_modules": 0, "packed_bytes": 0, "freed_bytes": 0}
    with torch.no_grad():
        # ---- twin VAE
        if dedup_vae and getattr(engine, "conditioner", None) is not None:
            for emb in engine.conditioner.embedders:
                if isinstance(emb, LatentEncoder) and emb.model is not engine.first_stage_model \
                        and _same_weights(emb.model, engine.first_stage_model):
                    emb.model = engine.first_stage_model
                    report["vae_deduplicated"] += 1
                    engine._vae_alias_prefixes = ("first_stage_model.", "conditioner.embedders.%d.model." % list(engine.conditioner.embedders).index(emb))
        # ---- pack roots (a fused child is packed through its parent only)
        fused = set()
        for m in engine.modules():
            if isinstance(m, H._Packed):
                fused.update(id(c) for c in m.fused_children())
        unet = engine.model.diffusion_model
        fused.update(id(rb.emb_layers[1]) for rb in unet._resblocks)
        roots = [m for m in engine.modules() if isinstance(m, H._Packed) and id(m) not in fused]
        # where the LayerNorm-folded layouts serve the forward pass (BasicTransformerBlock: fold = LN_GEMM and not fp8), the plain
        # q|k|v and GEGLU layouts of the UNet's blocks are never read: not packed (~270 MB of bf16 for the SD-2 UNet)
        from sgm.modules.attention import BasicTransformerBlock as _BTB
        fold_on = H.LN_GEMM
        ln_only = set()
        if fold_on:
            for m in unet.modules():
                if isinstance(m, _BTB):
                    ln_only.update((id(m.attn1), id(m.ff.net[0])))
            roots = [m for m in roots if id(m) not in ln_only]
        seen = set()
        for m in roots:
            if id(m) in seen:
                continue
            seen.add(id(m))
            pk = m.packed()
            report["packed_modules"] += 1
            for t in (pk if isinstance(pk, (tuple, list)) else (pk,)):
                if isinstance(t, torch.Tensor):
                    report["packed_bytes"] += t.numel() * t.element_size()
        w, b = unet._emb_pack()
        report["packed_bytes"] += w.numel() * w.element_size() + b.numel() * b.element_size()
        # ---- LayerNorm-folded layouts of the q|k|v and GEGLU projections (udt_ln_gemm_fwd)
        from sgm.modules.attention import BasicTransformerBlock
        # (in config #5 — UDT_FP8=1 — prepare_ln / prepare_mx8 also build, and freeze, the e4m3 layouts of the blocks whose linears
        #  run on MX8 operands: ahead of the release of the masters)
        if H.LN_GEMM:
            from sgm.modules.attention import SpatialTransformer
            for m in unet.modules():
                if isinstance(m, BasicTransformerBlock):
                    report["packed_bytes"] += m.prepare_ln(freeze=free_masters)
                elif isinstance(m, SpatialTransformer):
                    report["packed_bytes"] += m.prepare_mx8(freeze=free_masters)
                    report["packed_bytes"] += m.prepare_ffproj(freeze=free_masters)      # (round 6: [W_po W_2 | W_po], attention.FF_PROJ)
        # ---- release the masters
        if free_masters:
            victims = []
            for m in engine.modules():
                if isinstance(m, H._Packed) and (id(m) in seen or id(m) in fused or (id(m) in ln_only and getattr(m, "_pkln_frozen", False))):
                    m._pk_frozen = True
                    victims += list(m.parameters(recurse=False))
                    if hasattr(m, "own_masters"):
                        victims += m.own_masters()
            unet._emb_frozen = True
            done = set()
            for p in victims:
                if id(p) in done:
                    continue
                done.add(id(p))
                report["freed_bytes"] += p.numel() * p.element_size()
                p.data = torch.empty(0, dtype=p.dtype, device=p.device)
    return report
