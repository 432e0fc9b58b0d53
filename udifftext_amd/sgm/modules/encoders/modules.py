"""Conditioner and embedders of the UDiffText inference path on the gfx950 kernels.

Reference: sgm/modules/encoders/modules.py — AbstractEmbModel :48-102, GeneralConditioner :105-217,
SpatialRescaler :800-860, LatentEncoder :999-1014, PositionalEncoding :1069-1085, LabelEncoder :1088-1173.
The other embedders of that file (CLIP/T5/ViT...) are not referenced by configs/test/*.yaml and are out of scope;
this module deliberately imports none of kornia / open_clip / timm / transformers / torchvision / lightning.

Exact shortcuts taken: an embedder whose output is force-zeroed is not evaluated
(``zero_embedding``), and the masked-image VAE encoder runs once for c and uc when both see the same image —
only the posterior noise differs, drawn on the CPU in the reference's order (c first, then uc).
"""
from __future__ import annotations

import math
import string
from contextlib import nullcontext
from typing import Dict, List, Optional, Union

import torch
import torch.nn as nn

from udifftext_amd import ops

from ...util import count_params, disabled_train, expand_dims_like, instantiate_from_config, require_gpu
from .. import hipnn as H


class AbstractEmbModel(nn.Module):
    def __init__(self):
        super().__init__()
        self._is_trainable = None
        self._ucg_rate = None
        self._input_key = None
        self._emb_key = None

    is_trainable = property(lambda s: s._is_trainable, lambda s, v: setattr(s, "_is_trainable", v))
    ucg_rate = property(lambda s: s._ucg_rate, lambda s, v: setattr(s, "_ucg_rate", v))
    input_key = property(lambda s: s._input_key, lambda s, v: setattr(s, "_input_key", v))
    emb_key = property(lambda s: s._emb_key, lambda s, v: setattr(s, "_emb_key", v))


class GeneralConditioner(nn.Module):
    OUTPUT_DIM2KEYS = {2: "vector", 3: "crossattn", 4: "concat", 5: "concat"}
    KEY2CATDIM = {"vector": 1, "crossattn": 2, "concat": 1}

    def __init__(self, emb_models):
        super().__init__()
        embedders = []
        for n, cfg in enumerate(emb_models):
            emb = instantiate_from_config(cfg)
            assert isinstance(emb, AbstractEmbModel), \
                f"embedder model {emb.__class__.__name__} has to inherit from AbstractEmbModel"
            emb.is_trainable = cfg.get("is_trainable", False)
            emb.ucg_rate = cfg.get("ucg_rate", 0.0)
            if not emb.is_trainable:
                # NOTE: the reference installs ``disabled_train`` on a module still in training mode, which leaves
                # LabelEncoder's dropout active at inference (DESIGN.md "reference quirks"); here frozen embedders
                # are put in eval mode first, i.e. the deterministic network.
                emb.eval()
                emb.train = disabled_train
                emb.freeze()
            print(f"Initialized embedder #{n}: {emb.__class__.__name__} with {count_params(emb, False)} params. "
                  f"Trainable: {emb.is_trainable}")
            if "emb_key" in cfg:
                emb.emb_key = cfg["emb_key"]
            if "input_key" in cfg:
                emb.input_key = cfg["input_key"]
            elif "input_keys" in cfg:
                emb.input_keys = cfg["input_keys"]
            else:
                raise KeyError(f"need either 'input_key' or 'input_keys' for embedder {emb.__class__.__name__}")
            if cfg.get("legacy_ucg_value", None) is not None:
                raise NotImplementedError("legacy_ucg_value is a training-time feature")
            emb.legacy_ucg_val = None
            embedders.append(emb)
        self.embedders = nn.ModuleList(embedders)

    def forward(self, batch: Dict, force_zero_embeddings: Optional[List] = None) -> Dict:
        output = {}
        force_zero_embeddings = force_zero_embeddings or []
        for emb in self.embedders:
            zeroed = getattr(emb, "input_key", None) is not None and emb.input_key in force_zero_embeddings
            with (nullcontext() if emb.is_trainable else torch.no_grad()):
                if getattr(emb, "input_key", None) is not None:
                    arg = batch[emb.input_key]
                    if zeroed and hasattr(emb, "zero_embedding"):
                        out = emb.zero_embedding(arg)
                    else:
                        out = emb(arg)
                else:
                    out = emb(*[batch[k] for k in emb.input_keys])
            assert isinstance(out, (torch.Tensor, list, tuple)), \
                f"encoder outputs must be tensors or a sequence, but got {type(out)}"
            for e in (out if isinstance(out, (list, tuple)) else [out]):
                key = emb.emb_key if emb.emb_key is not None else self.OUTPUT_DIM2KEYS[e.dim()]
                if emb.ucg_rate > 0.0:
                    keep = torch.bernoulli((1.0 - emb.ucg_rate) * torch.ones(e.shape[0], device=e.device))
                    e = expand_dims_like(keep, e) * e
                if zeroed:
                    e = torch.zeros_like(e)
                    e._udt_all_zero = True           # known on the host: consumers need no device round trip to find out
                output[key] = torch.cat((output[key], e), self.KEY2CATDIM[key]) if key in output else e
        return output

    def get_unconditional_conditioning(self, batch_c, batch_uc=None, force_uc_zero_embeddings=None):
        force_uc_zero_embeddings = force_uc_zero_embeddings or []
        rates = [e.ucg_rate for e in self.embedders]
        for e in self.embedders:
            e.ucg_rate = 0.0
        buc = batch_c if batch_uc is None else batch_uc
        # embedders that may reuse their deterministic part when c and uc feed them the same tensor
        for e in self.embedders:
            if hasattr(e, "share_between_calls"):
                k = e.input_key
                # (pipeline.prepare_batch marks its unconditional batch as a clone of the conditional one: no device-side
                #  comparison — a host sync that would stall the launch thread while other batches are in flight)
                st = buc.get("_udt_clone_state", {}).get(k) if isinstance(buc, dict) else None
                marked = (buc.get("_udt_clone_of") is batch_c and k not in buc.get("_udt_changed", ()) and st is not None
                          and st == (id(buc[k]), buc[k]._version, id(batch_c[k]), batch_c[k]._version))   # neither side touched since
                same = (buc[k] is batch_c[k]) or marked \
                    or (buc[k].shape == batch_c[k].shape and bool(torch.equal(buc[k], batch_c[k])))
                e.share_between_calls(same)
        try:
            c = self(batch_c)
            uc = self(buc, force_uc_zero_embeddings)
        finally:
            for e, r in zip(self.embedders, rates):
                e.ucg_rate = r
                if hasattr(e, "share_between_calls"):
                    e.share_between_calls(False)
        return c, uc


class SpatialRescaler(AbstractEmbModel):
    """mask [B,1,H,W] -> [B,1,H/8,W/8]: F.interpolate(scale 0.125, bilinear, align_corners False) == mean of the
    centre 2x2 pixels of each 8x8 block (verified in tests/test_oracle_golden.py::test_g4_mask_rescale)."""

    def __init__(self, n_stages=1, method="bilinear", multiplier=0.5, in_channels=3, out_channels=None, bias=False,
                 wrap_video=False, kernel_size=1, remap_output=False):
        super().__init__()
        if n_stages != 1 or method != "bilinear" or multiplier != 0.125 or in_channels != 1 or out_channels is not None \
                or wrap_video or remap_output:
            raise NotImplementedError("SpatialRescaler is implemented for the UDiffText mask path "
                                      "(in_channels 1, bilinear x0.125, one stage)")
        self.multiplier = multiplier

    def freeze(self):
        pass

    def forward(self, x):
        require_gpu(x, "SpatialRescaler")
        return ops.mask_downsample(x.float().contiguous())

    def encode(self, x):
        return self(x)


class LatentEncoder(AbstractEmbModel):
    def __init__(self, scale_factor, config, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.scale_factor = scale_factor
        self.model = instantiate_from_config(config).eval()
        self.model.train = disabled_train
        self._share = False
        self._moments = None

    def freeze(self):
        for p in self.model.parameters():
            p.requires_grad = False

    def share_between_calls(self, on: bool):
        self._share = on
        self._moments = None

    def forward(self, x):
        from ..distributions.distributions import DiagonalGaussianDistribution
        if self._share and self._moments is not None:
            mom = self._moments
        else:
            mom = self.model.encode_moments(x)
            if self._share:
                self._moments = mom
        # scale_factor * (mean + std * eps): the scale is folded into the sampling kernel
        return DiagonalGaussianDistribution(mom).sample(scale=self.scale_factor)


class PositionalEncoding(nn.Module):
    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        pe = torch.zeros(max_len, d_model)
        pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        self.register_buffer("pe", pe)


class _SelfAttnParams(nn.Module):
    """parameter names of nn.MultiheadAttention: in_proj_weight / in_proj_bias / out_proj.{weight,bias}"""

    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        H._init_uniform_(self.in_proj_weight, d)
        self.out_proj = H.Linear(d, d)


class _EncoderLayer(H._Packed):
    """post-norm nn.TransformerEncoderLayer(d_model, nhead, dim_feedforward=2048, relu, batch_first)"""

    def __init__(self, d, heads, ff):
        super().__init__()
        self.d, self.heads = d, heads
        self.self_attn = _SelfAttnParams(d)
        self.linear1 = H.Linear(d, ff)
        self.linear2 = H.Linear(ff, d)
        self.norm1 = H.LayerNorm(d)
        self.norm2 = H.LayerNorm(d)

    def _key(self):
        p = self.self_attn.in_proj_weight
        return ((p.data_ptr(), p._version, str(p.device)),)

    def own_masters(self):
        """parameters consumed only through this module's pack (not Linear / Conv2d children)"""
        return [self.self_attn.in_proj_weight, self.self_attn.in_proj_bias]

    def _pack(self):
        from udifftext_amd import packing
        return packing.pack_linear(self.self_attn.in_proj_weight), packing.pad_bias(self.self_attn.in_proj_bias)   # (a COPY: own_masters() are released)

    def forward(self, x, B, Lc):
        d, Hh = self.d, self.heads
        w, b = self.packed()
        qkv = ops.linear(x, w, b).reshape(B, Lc, 3 * d)
        a = ops.xattention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], Hh, d // Hh, (d // Hh) ** -0.5)
        x = self.norm1(self.self_attn.out_proj(a.reshape(B * Lc, d), residual=x))
        f = self.linear1(x, flags=H.GEMM_RELU)
        return self.norm2(self.linear2(f, residual=x))


class _Encoder(nn.Module):
    def __init__(self, d, heads, ff, n):
        super().__init__()
        self.layers = nn.ModuleList([_EncoderLayer(d, heads, ff) for _ in range(n)])


class _Embedding(nn.Module):
    def __init__(self, n, d):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, d))
        if not H.init_skipped():
            nn.init.normal_(self.weight)


class LabelEncoder(AbstractEmbModel):
    """characters -> [B, max_len, emb_dim]: Embedding(95, d) + sinusoidal PE -> 12 post-norm encoder layers,
    no padding mask (index 0 = padding = unknown character)."""

    def __init__(self, max_len, emb_dim, n_heads=8, n_trans_layers=12, ckpt_path=None, trainable=False, lr=1e-4,
                 lambda_cls=0.1, lambda_pos=0.1, clip_dim=1024, visual_len=197, visual_dim=768, visual_config=None,
                 *args, **kwargs):
        super().__init__(*args, **kwargs)
        if trainable:
            raise NotImplementedError("LabelEncoder pre-training heads are out of scope (inference path only)")
        self.max_len, self.emd_dim = max_len, emb_dim
        self.n_heads, self.n_trans_layers = n_heads, n_trans_layers
        self.character = string.printable[:-6]
        self.num_cls = len(self.character) + 1
        self.label_embedding = _Embedding(self.num_cls, emb_dim)
        self.pos_embedding = PositionalEncoding(d_model=emb_dim, max_len=max_len)
        self.encoder = _Encoder(emb_dim, n_heads, 2048, n_trans_layers)
        if ckpt_path is not None:
            self.load_state_dict(torch.load(ckpt_path, map_location="cpu")["state_dict"], strict=False)

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False

    def get_index(self, labels):
        rows = []
        for label in labels:
            assert len(label) <= self.max_len
            idx = [self.character.find(c) + 1 for c in label]
            rows.append(idx + [0] * (self.max_len - len(idx)))
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            return torch.tensor(rows, device=dev)
        return torch.tensor(rows).pin_memory().to(dev, non_blocking=True)     # (a pageable copy would block the launch thread)

    def get_embeddings(self, x):
        require_gpu(x, "LabelEncoder")
        B, Lc = x.shape
        h = ops.embed_tokens(x.to(torch.int32).contiguous().reshape(-1), self.label_embedding.weight,
                             self.pos_embedding.pe)
        for layer in self.encoder.layers:
            h = layer(h, B, Lc)
        return h.reshape(B, Lc, self.emd_dim).float()

    def zero_embedding(self, labels):
        dev = next(self.parameters()).device
        return torch.zeros((len(labels), self.max_len, self.emd_dim), dtype=torch.float32, device=dev)

    def forward(self, labels):
        return self.get_embeddings(self.get_index(labels))
