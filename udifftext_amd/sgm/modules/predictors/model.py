"""OCR scorer on the gfx950 kernels: ``ParseqPredictor`` (reference sgm/modules/predictors/model.py:7-57) and the PARSeq
network it wraps (reference src/parseq/strhub/models/parseq/system.py:36-138, modules.py:27-126; hub entry ``parseq``:
src/parseq/hubconf.py:20-27, configs/model/parseq.yaml).  It runs AFTER the denoising path (test.py:74-91: crops of the
decoded frames -> text) and inside the training loss (loss.py:188) — SURVEY.md §8f-2.

State-dict names under ``self.parseq`` are the reference checkpoint's (``parseq-bb5792a6.pt`` is loaded INTO
``self.parseq``, model.py:12): ``encoder.*`` as timm's VisionTransformer names them, ``decoder.layers.0.*``,
``head``, ``text_embed.embedding``, ``pos_queries``.

Every matrix product, LayerNorm and attention below is a launch into libudt_kernels (bf16 storage, fp32 accumulation and
statistics; logits in fp32); there is no CPU path.  torch is used for gathers, concatenation and the argmax of the
autoregressive loop, and for torchvision's ``Resize(BICUBIC, antialias=True)`` (= ``F.interpolate(..., antialias=True)``,
an ATen kernel: 3x32x128 pixels per crop).  The exact-erf GELU of the MLPs rides on the GEGLU epilogue with a constant
value branch (weight 0, bias 1): ``1 * gelu(x W^T + b)``.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from udifftext_amd import ops, packing

from .. import hipnn as H

CHARSET_94 = "0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~"


class Tokenizer:
    """strhub/data/utils.py:106-139: ids = [EOS] + charset + [BOS, PAD]; greedy decode truncated at the first EOS"""
    BOS, EOS, PAD = "[B]", "[E]", "[P]"

    def __init__(self, charset: str = CHARSET_94):
        self._itos = (self.EOS,) + tuple(charset) + (self.BOS, self.PAD)
        self._stoi = {s: i for i, s in enumerate(self._itos)}
        self.eos_id, self.bos_id, self.pad_id = self._stoi[self.EOS], self._stoi[self.BOS], self._stoi[self.PAD]

    def __len__(self):
        return len(self._itos)

    def encode(self, labels: Sequence[str], device=None) -> torch.Tensor:
        rows = [[self.bos_id] + [self._stoi[c] for c in y] + [self.eos_id] for y in labels]
        n = max(len(r) for r in rows)
        return torch.tensor([r + [self.pad_id] * (n - len(r)) for r in rows], dtype=torch.long, device=device)

    def decode(self, token_dists: torch.Tensor) -> Tuple[List[str], List[torch.Tensor]]:
        labels, probs = [], []
        for dist in token_dists:
            p, ids = dist.max(-1)
            ids = ids.tolist()
            cut = ids.index(self.eos_id) if self.eos_id in ids else len(ids)
            labels.append("".join(self._itos[i] for i in ids[:cut]))
            probs.append(p[:cut + 1])
        return labels, probs


class _GeluLinear(H._Packed):
    """Linear + exact GELU in one GEMM launch (GEGLU epilogue with the value branch held at 1)"""

    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        H._init_uniform_(self.weight, in_features)
        H._init_uniform_(self.bias, in_features)

    def _pack(self):
        w = torch.cat([torch.zeros_like(self.weight), self.weight], 0)
        b = torch.cat([torch.ones_like(self.bias), self.bias], 0)
        return packing.pack_geglu(w, b)

    def forward(self, x):
        w, b = self.packed()
        return ops.linear(x, w, b, flags=H.GEMM_GEGLU)


class _MHA(H._Packed):
    """nn.MultiheadAttention(batch_first=True) with the packed in_proj (modules.py:35-36)"""

    def __init__(self, dim: int, heads: int):
        super().__init__()
        self.dim, self.heads = dim, heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * dim, dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * dim))
        H._init_uniform_(self.in_proj_weight, dim)
        self.out_proj = H.Linear(dim, dim)

    def _pack(self):
        C = self.dim
        w, b = self.in_proj_weight, self.in_proj_bias
        return (packing.pack_linear(w[:C]), b[:C].float().clone(),
                packing.pack_linear(w[C:]), b[C:].float().clone())

    def project_kv(self, kv):
        """k|v rows of a key/value source [B, L, C] -> [B, L, 2C] (constant for the memory during decoding)"""
        _, _, wkv, bkv = self.packed()
        B, Lk, C = kv.shape
        return ops.linear(kv.reshape(B * Lk, C), wkv, bkv).reshape(B, Lk, 2 * C)

    def forward(self, q_in, kv=None, kv_proj=None, attn_mask=None, key_padding_mask=None, residual=None):
        wq, bq, _, _ = self.packed()
        B, Lq, C = q_in.shape
        q = ops.linear(q_in.reshape(B * Lq, C), wq, bq).reshape(B, Lq, C)
        if kv_proj is None:
            kv_proj = self.project_kv(kv)
        o = ops.masked_attention(q, kv_proj[..., :C], kv_proj[..., C:], self.heads, (C // self.heads) ** -0.5,
                                 mask=attn_mask, key_padding_mask=key_padding_mask)
        res = None if residual is None else residual.reshape(B * Lq, C)
        return self.out_proj(o.reshape(B * Lq, C), residual=res).reshape(B, Lq, C)


# ------------------------------------------------------------------------------------------------ encoder (ViT)
class _PatchEmbed(H._Packed):
    def __init__(self, patch_size, in_chans, embed_dim):
        super().__init__()
        self.patch_size = tuple(patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)   # parameter holder

    def _key(self):
        return tuple((p.data_ptr(), p._version, str(p.device)) for p in self.proj.parameters())

    def _pack(self):
        w = self.proj.weight
        return packing.pack_linear(w.reshape(w.shape[0], -1)), packing.pad_bias(self.proj.bias)   # (a copy: never alias a master)

    def forward(self, img):
        """img fp32 [B, 3, H, W] -> tokens bf16 [B, (H/ph)*(W/pw), E]; token order y*W' + x as Conv2d + flatten(2)"""
        B, Cin, Hh, Ww = img.shape
        ph, pw = self.patch_size
        w, b = self.packed()
        cols = img.reshape(B, Cin, Hh // ph, ph, Ww // pw, pw).permute(0, 2, 4, 1, 3, 5).reshape(-1, Cin * ph * pw)
        a = torch.zeros((cols.shape[0], w.shape[1]), dtype=torch.bfloat16, device=img.device)
        a[:, :cols.shape[1]] = cols.to(torch.bfloat16)
        return ops.linear(a, w, b).reshape(B, (Hh // ph) * (Ww // pw), -1)


class _VitAttention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.heads = heads
        self.qkv = H.Linear(dim, 3 * dim)
        self.proj = H.Linear(dim, dim)

    def forward(self, xn, residual):
        B, N, C = xn.shape
        qkv = self.qkv(xn.reshape(B * N, C)).reshape(B, N, 3 * C)
        o = ops.masked_attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], self.heads, (C // self.heads) ** -0.5)
        return self.proj(o.reshape(B * N, C), residual=residual.reshape(B * N, C)).reshape(B, N, C)


class _VitMlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = _GeluLinear(dim, hidden)
        self.fc2 = H.Linear(hidden, dim)

    def forward(self, xn, residual):
        B, N, C = xn.shape
        return self.fc2(self.fc1(xn.reshape(B * N, C)), residual=residual.reshape(B * N, C)).reshape(B, N, C)


class _VitBlock(nn.Module):
    def __init__(self, dim, heads, mlp_ratio):
        super().__init__()
        self.norm1 = H.LayerNorm(dim, eps=1e-6)
        self.attn = _VitAttention(dim, heads)
        self.norm2 = H.LayerNorm(dim, eps=1e-6)
        self.mlp = _VitMlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        x = self.attn(self.norm1(x), x)
        return self.mlp(self.norm2(x), x)


class Encoder(nn.Module):
    """timm~=0.6.5 ``VisionTransformer(class_token=False, num_classes=0, global_pool='')`` as modules.py:99-110 builds it:
    forward_features returns every token after the final LayerNorm"""

    def __init__(self, img_size, patch_size, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.):
        super().__init__()
        n_patches = (img_size[0] // patch_size[0]) * (img_size[1] // patch_size[1])
        self.patch_embed = _PatchEmbed(patch_size, in_chans, embed_dim)
        self.pos_embed = nn.Parameter(torch.zeros(1, n_patches, embed_dim))
        self.blocks = nn.Sequential(*[_VitBlock(embed_dim, num_heads, mlp_ratio) for _ in range(depth)])
        self.norm = H.LayerNorm(embed_dim, eps=1e-6)

    def forward(self, img):
        x = self.patch_embed(img)
        x = (x.float() + self.pos_embed).to(torch.bfloat16)
        return self.norm(self.blocks(x))


# ------------------------------------------------------------------------------------------------ decoder
class DecoderLayer(nn.Module):
    """pre-LN two-stream decoder layer (modules.py:27-86); dropout is inactive at inference"""

    def __init__(self, d_model, nhead, dim_feedforward):
        super().__init__()
        self.self_attn = _MHA(d_model, nhead)
        self.cross_attn = _MHA(d_model, nhead)
        self.linear1 = _GeluLinear(d_model, dim_feedforward)
        self.linear2 = H.Linear(dim_feedforward, d_model)
        self.norm1 = H.LayerNorm(d_model)
        self.norm2 = H.LayerNorm(d_model)
        self.norm_q = H.LayerNorm(d_model)
        self.norm_c = H.LayerNorm(d_model)

    def forward_stream(self, tgt, tgt_norm, tgt_kv, memory_kv, tgt_mask, tgt_key_padding_mask):
        tgt = self.self_attn(tgt_norm, kv=tgt_kv, attn_mask=tgt_mask, key_padding_mask=tgt_key_padding_mask, residual=tgt)
        tgt = self.cross_attn(self.norm1(tgt), kv_proj=memory_kv, residual=tgt)
        B, Lq, C = tgt.shape
        h = self.linear1(self.norm2(tgt).reshape(B * Lq, C))
        return self.linear2(h, residual=tgt.reshape(B * Lq, C)).reshape(B, Lq, C)

    def forward(self, query, content, memory_kv, query_mask=None, content_mask=None, content_key_padding_mask=None,
                update_content=True):
        query_norm = self.norm_q(query)
        content_norm = self.norm_c(content)
        query = self.forward_stream(query, query_norm, content_norm, memory_kv, query_mask, content_key_padding_mask)
        if update_content:
            content = self.forward_stream(content, content_norm, content_norm, memory_kv, content_mask,
                                          content_key_padding_mask)
        return query, content


class Decoder(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([DecoderLayer(d_model, nhead, dim_feedforward) for _ in range(num_layers)])
        self.norm = H.LayerNorm(d_model)

    def memory_kv(self, memory):
        """the cross-attention k|v projections of the encoder output: constant over the decoding steps"""
        return [layer.cross_attn.project_kv(memory) for layer in self.layers]

    def forward(self, query, content, memory_kv, query_mask=None, content_mask=None, content_key_padding_mask=None):
        for i, mod in enumerate(self.layers):
            last = i == len(self.layers) - 1
            query, content = mod(query, content, memory_kv[i], query_mask, content_mask, content_key_padding_mask,
                                 update_content=not last)
        return self.norm(query)


class TokenEmbedding(nn.Module):
    def __init__(self, charset_size, embed_dim):
        super().__init__()
        self.embedding = nn.Embedding(charset_size, embed_dim)
        self.embed_dim = embed_dim

    def forward(self, tokens):
        return math.sqrt(self.embed_dim) * self.embedding(tokens)


class PARSeq(nn.Module):
    """inference side of strhub.models.parseq.system.PARSeq (system.py:36-138) with the ``parseq`` hub hyper-parameters"""

    def __init__(self, charset: str = CHARSET_94, max_label_length: int = 25, img_size=(32, 128), patch_size=(4, 8),
                 embed_dim: int = 384, enc_num_heads: int = 6, enc_mlp_ratio: int = 4, enc_depth: int = 12,
                 dec_num_heads: int = 12, dec_mlp_ratio: int = 4, dec_depth: int = 1, decode_ar: bool = True,
                 refine_iters: int = 1):
        super().__init__()
        self.tokenizer = Tokenizer(charset)
        self.bos_id, self.eos_id, self.pad_id = self.tokenizer.bos_id, self.tokenizer.eos_id, self.tokenizer.pad_id
        self.max_label_length, self.decode_ar, self.refine_iters = max_label_length, decode_ar, refine_iters
        self.hparams = type("HP", (), {"img_size": tuple(img_size)})()            # predictors/model.py:15 reads it
        self.pos_queries = nn.Parameter(torch.zeros(1, max_label_length + 1, embed_dim))
        self.encoder = Encoder(img_size, patch_size, embed_dim=embed_dim, depth=enc_depth, num_heads=enc_num_heads,
                               mlp_ratio=enc_mlp_ratio)
        self.decoder = Decoder(embed_dim, dec_num_heads, embed_dim * dec_mlp_ratio, dec_depth)
        self.head = H.Linear(embed_dim, len(self.tokenizer) - 2)
        self.text_embed = TokenEmbedding(len(self.tokenizer), embed_dim)
        if not H.init_skipped():
            nn.init.trunc_normal_(self.pos_queries, std=.02)

    def encode(self, img):
        return self.encoder(img)

    def decode(self, tgt, memory_kv, tgt_mask=None, tgt_padding_mask=None, tgt_query=None, tgt_query_mask=None):
        N, L = tgt.shape
        null_ctx = self.text_embed(tgt[:, :1])
        tgt_emb = torch.cat([null_ctx, self.pos_queries[:, :L - 1] + self.text_embed(tgt[:, 1:])], dim=1)
        if tgt_query is None:
            tgt_query = self.pos_queries[:, :L].expand(N, -1, -1)
        q = tgt_query.to(torch.bfloat16).contiguous()
        return self.decoder(q, tgt_emb.to(torch.bfloat16).contiguous(), memory_kv, tgt_query_mask, tgt_mask, tgt_padding_mask)

    def logits_of(self, out):
        B, Lq, C = out.shape
        return self.head(out.reshape(B * Lq, C), flags=H.GEMM_OUT_F32)[:, :len(self.tokenizer) - 2].reshape(B, Lq, -1)

    @torch.no_grad()
    def forward(self, images, max_length: Optional[int] = None, memory=None):
        """images fp32 [N, 3, 32, 128] on the GPU -> fp32 logits [N, L, 95]; L <= max_label_length + 1 (the AR loop stops
        once every sample has emitted EOS, system.py:122-124)"""
        dev = images.device
        testing = max_length is None
        max_length = self.max_label_length if max_length is None else min(max_length, self.max_label_length)
        bs = images.shape[0]
        num_steps = max_length + 1
        if memory is None:
            memory = self.encode(images)
        memory_kv = self.decoder.memory_kv(memory)
        pos_queries = self.pos_queries[:, :num_steps].expand(bs, -1, -1)
        tgt_mask = query_mask = torch.triu(torch.full((num_steps, num_steps), float("-inf"), device=dev), 1)
        if self.decode_ar:
            tgt_in = torch.full((bs, num_steps), self.pad_id, dtype=torch.long, device=dev)
            tgt_in[:, 0] = self.bos_id
            logits = []
            for i in range(num_steps):
                j = i + 1
                out = self.decode(tgt_in[:, :j], memory_kv, tgt_mask[:j, :j], tgt_query=pos_queries[:, i:j],
                                  tgt_query_mask=query_mask[i:j, :j])
                p_i = self.logits_of(out)
                logits.append(p_i)
                if j < num_steps:
                    tgt_in[:, j] = p_i.squeeze(1).argmax(-1)
                    if testing and bool((tgt_in == self.eos_id).any(dim=-1).all()):
                        break
            logits = torch.cat(logits, dim=1)
            self.last_ar_tokens = tgt_in[:, :logits.shape[1]].clone()      # (diagnostics / tests: the greedy prefix)
        else:
            tgt_in = torch.full((bs, 1), self.bos_id, dtype=torch.long, device=dev)
            logits = self.logits_of(self.decode(tgt_in, memory_kv, tgt_query=pos_queries))
        if self.refine_iters:
            query_mask = query_mask.clone()
            query_mask[torch.triu(torch.ones(num_steps, num_steps, dtype=torch.bool, device=dev), 2)] = 0
            bos = torch.full((bs, 1), self.bos_id, dtype=torch.long, device=dev)
            for _ in range(self.refine_iters):
                tgt_in = torch.cat([bos, logits[:, :-1].argmax(-1)], dim=1)
                kpm = (tgt_in == self.eos_id).int().cumsum(-1) > 0
                Lt = tgt_in.shape[1]
                out = self.decode(tgt_in, memory_kv, tgt_mask[:Lt, :Lt], kpm, tgt_query=pos_queries[:, :Lt],
                                  tgt_query_mask=query_mask[:Lt, :Lt])
                logits = self.logits_of(out)
        return logits


class ParseqPredictor(nn.Module):
    """reference sgm/modules/predictors/model.py:7-57 — same constructor, ``forward`` (list of [3,h,w] crops in [0,1] ->
    logits), ``img2txt`` and ``calc_loss``"""

    def __init__(self, ckpt_path=None, freeze=True, *args, **kwargs):
        super().__init__()
        self.parseq = PARSeq().eval()
        if ckpt_path is not None:
            self.parseq.load_state_dict(torch.load(ckpt_path, map_location="cpu"))
        if freeze:
            self.freeze()

    def freeze(self):
        for p in self.parseq.parameters():
            p.requires_grad_(False)

    def transform(self, x):
        size = self.parseq.hparams.img_size
        dev = next(self.parameters()).device
        x = torch.cat([F.interpolate(t[None].float().to(dev), size=size, mode="bicubic", antialias=True, align_corners=False)
                       for t in x])
        return (x - 0.5) / 0.5

    def forward(self, x):
        return self.parseq(self.transform(x))

    def img2txt(self, x):
        label, _ = self.parseq.tokenizer.decode(self(x))
        return label

    def calc_loss(self, x, label):
        preds = self(x)
        gt_ids = self.parseq.tokenizer.encode(label).to(preds.device)
        losses = []
        for pred, gt_id in zip(preds, gt_ids):
            eos_id = (gt_id == 0).nonzero().item()
            ce = F.cross_entropy(pred[:eos_id - 1].permute(1, 0)[None], gt_id[1:eos_id][None])
            losses.append(torch.clamp(ce, max=1.0)[None])
        return torch.cat(losses)
