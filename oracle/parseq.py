"""TEST INFRASTRUCTURE — fp32 CPU restatement of the OCR scorer: ``ParseqPredictor``
(reference sgm/modules/predictors/model.py:7-57) = torchvision resize + normalise + PARSeq
(reference src/parseq/strhub/models/parseq/system.py:36-138, modules.py:27-126) + the tokenizer
(src/parseq/strhub/data/utils.py:45-139).  Only tests/ may import this; the product is
udifftext_amd/sgm/modules/predictors/model.py on the HIP kernels.

Pinning.  The decoder, the autoregressive + refinement decoding loop, the token embedding, the head and the
tokenizer are pinned by tests/golden/parseq_golden.npz, produced by running the REAL reference classes
(strhub.models.parseq.system.PARSeq / modules.Decoder) in the build container (tests/golden/make_golden.py
--parseq).  The encoder is ``timm.models.vision_transformer.VisionTransformer`` (timm~=0.6.5,
src/parseq/requirements.txt:4), a third-party dependency that is NOT installed here and not vendored by the
reference: ``vit_encode`` restates its published algorithm (patch-embedding conv -> + pos_embed -> pre-LN blocks
with eps 1e-6, qkv bias, exact-erf GELU MLP -> final LayerNorm; class_token=False, num_classes=0 as
modules.py:99-110 constructs it) — **parity unpinned** for that part, and for torchvision's
``Resize(BICUBIC, antialias=True)`` (= ``F.interpolate(mode="bicubic", antialias=True, align_corners=False)``).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

CHARSET_94 = "0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~"
# hyper-parameters of the `parseq` hub entry (src/parseq/configs/model/parseq.yaml, configs/main.yaml)
HP = dict(img_size=(32, 128), patch_size=(4, 8), embed_dim=384, enc_num_heads=6, enc_mlp_ratio=4, enc_depth=12,
          dec_num_heads=12, dec_mlp_ratio=4, dec_depth=1, max_label_length=25, decode_ar=True, refine_iters=1)


# ---------------------------------------------------------------------------------------------- tokenizer
class Tokenizer:
    """strhub/data/utils.py:106-139: ids = [EOS] + charset + [BOS, PAD]"""

    def __init__(self, charset: str = CHARSET_94):
        self.itos = ("[E]",) + tuple(charset) + ("[B]", "[P]")
        self.stoi = {s: i for i, s in enumerate(self.itos)}
        self.eos_id, self.bos_id, self.pad_id = 0, len(charset) + 1, len(charset) + 2

    def __len__(self):
        return len(self.itos)

    def encode(self, labels: Sequence[str]) -> torch.Tensor:
        rows = [[self.bos_id] + [self.stoi[c] for c in y] + [self.eos_id] for y in labels]
        n = max(len(r) for r in rows)
        return torch.tensor([r + [self.pad_id] * (n - len(r)) for r in rows], dtype=torch.long)

    def decode(self, dists: torch.Tensor) -> Tuple[List[str], List[torch.Tensor]]:
        """greedy; truncate at the first EOS (its probability is kept) — utils.py:84-104,128-139"""
        labels, probs = [], []
        for dist in dists:
            p, ids = dist.max(-1)
            ids = ids.tolist()
            cut = ids.index(self.eos_id) if self.eos_id in ids else len(ids)
            labels.append("".join(self.itos[i] for i in ids[:cut]))
            probs.append(p[:cut + 1])
        return labels, probs


# ------------------------------------------------------------------------------------------------ encoder
def _ln(x, sd, name, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def vit_encode(sd: Dict[str, torch.Tensor], img: torch.Tensor, heads: int = 6, prefix: str = "encoder.") -> torch.Tensor:
    """timm 0.6.x VisionTransformer.forward_features with class_token=False: [B,3,32,128] -> [B,128,384]"""
    x = F.conv2d(img, sd[prefix + "patch_embed.proj.weight"], sd[prefix + "patch_embed.proj.bias"],
                 stride=sd[prefix + "patch_embed.proj.weight"].shape[-2:])
    x = x.flatten(2).transpose(1, 2) + sd[prefix + "pos_embed"]
    B, N, C = x.shape
    D = C // heads
    i = 0
    while f"{prefix}blocks.{i}.norm1.weight" in sd:
        p = f"{prefix}blocks.{i}."
        h = _ln(x, sd, p + "norm1", 1e-6)
        qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(B, N, 3, heads, D).permute(2, 0, 3, 1, 4)
        a = torch.softmax(qkv[0] @ qkv[1].transpose(-2, -1) * D ** -0.5, dim=-1) @ qkv[2]
        x = x + F.linear(a.transpose(1, 2).reshape(B, N, C), sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        h = _ln(x, sd, p + "norm2", 1e-6)
        h = F.linear(F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"],
                     sd[p + "mlp.fc2.bias"])
        x = x + h
        i += 1
    return _ln(x, sd, prefix + "norm", 1e-6)


# ------------------------------------------------------------------------------------------------ decoder
def _mha(sd, name, q_in, kv_in, heads, attn_mask=None, key_padding_mask=None):
    """nn.MultiheadAttention(batch_first=True) forward, packed in_proj (modules.py:35-36)"""
    C = q_in.shape[-1]
    D = C // heads
    w, b = sd[name + ".in_proj_weight"], sd[name + ".in_proj_bias"]
    q = F.linear(q_in, w[:C], b[:C])
    k = F.linear(kv_in, w[C:2 * C], b[C:2 * C])
    v = F.linear(kv_in, w[2 * C:], b[2 * C:])
    B, Lq, Lk = q.shape[0], q.shape[1], k.shape[1]
    q = q.reshape(B, Lq, heads, D).transpose(1, 2)
    k = k.reshape(B, Lk, heads, D).transpose(1, 2)
    v = v.reshape(B, Lk, heads, D).transpose(1, 2)
    s = q @ k.transpose(-2, -1) * D ** -0.5
    if attn_mask is not None:
        s = s + attn_mask
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, Lq, C)
    return F.linear(o, sd[name + ".out_proj.weight"], sd[name + ".out_proj.bias"])


def decoder_layer_stream(sd, p, heads, tgt, tgt_norm, tgt_kv, memory, tgt_mask, tgt_kpm):
    """DecoderLayer.forward_stream (modules.py:57-74), dropout inactive"""
    tgt = tgt + _mha(sd, p + "self_attn", tgt_norm, tgt_kv, heads, tgt_mask, tgt_kpm)
    tgt = tgt + _mha(sd, p + "cross_attn", _ln(tgt, sd, p + "norm1", 1e-5), memory, heads)
    h = _ln(tgt, sd, p + "norm2", 1e-5)
    h = F.linear(F.gelu(F.linear(h, sd[p + "linear1.weight"], sd[p + "linear1.bias"])), sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    return tgt + h


def decoder(sd, query, content, memory, query_mask=None, content_mask=None, content_kpm=None, heads: int = 12,
            prefix: str = "decoder."):
    """Decoder.forward (modules.py:96-104): two-stream layers, the last one leaves the content stream alone"""
    n = 0
    while f"{prefix}layers.{n}.norm_q.weight" in sd:
        n += 1
    for i in range(n):
        p = f"{prefix}layers.{i}."
        qn = _ln(query, sd, p + "norm_q", 1e-5)
        cn = _ln(content, sd, p + "norm_c", 1e-5)
        query = decoder_layer_stream(sd, p, heads, query, qn, cn, memory, query_mask, content_kpm)
        if i != n - 1:
            content = decoder_layer_stream(sd, p, heads, content, cn, cn, memory, content_mask, content_kpm)
    return _ln(query, sd, prefix + "norm", 1e-5)


def decode(sd, tgt, memory, tgt_mask=None, tgt_kpm=None, tgt_query=None, tgt_query_mask=None, heads: int = 12):
    """PARSeq.decode (system.py:83-95)"""
    N, L = tgt.shape
    C = sd["pos_queries"].shape[-1]
    emb = lambda t: math.sqrt(C) * sd["text_embed.embedding.weight"][t]
    null_ctx = emb(tgt[:, :1])
    tgt_emb = torch.cat([null_ctx, sd["pos_queries"][:, :L - 1] + emb(tgt[:, 1:])], dim=1)
    if tgt_query is None:
        tgt_query = sd["pos_queries"][:, :L].expand(N, -1, -1)
    return decoder(sd, tgt_query, tgt_emb, memory, tgt_query_mask, tgt_mask, tgt_kpm, heads)


def parseq_forward(sd: Dict[str, torch.Tensor], images: torch.Tensor, max_length: Optional[int] = None,
                   memory: Optional[torch.Tensor] = None, enc_heads: int = 6, dec_heads: int = 12,
                   max_label_length: int = 25, decode_ar: bool = True, refine_iters: int = 1,
                   tok: Optional[Tokenizer] = None) -> torch.Tensor:
    """PARSeq.forward (system.py:97-138) -> logits [N, L, len(tokenizer) - 2]"""
    tok = tok or Tokenizer()
    testing = max_length is None
    max_length = max_label_length if max_length is None else min(max_length, max_label_length)
    bs = images.shape[0]
    num_steps = max_length + 1
    if memory is None:
        memory = vit_encode(sd, images, enc_heads)
    pos_queries = sd["pos_queries"][:, :num_steps].expand(bs, -1, -1)
    tgt_mask = query_mask = torch.triu(torch.full((num_steps, num_steps), float("-inf")), 1)
    head = lambda t: F.linear(t, sd["head.weight"], sd["head.bias"])
    if decode_ar:
        tgt_in = torch.full((bs, num_steps), tok.pad_id, dtype=torch.long)
        tgt_in[:, 0] = tok.bos_id
        logits = []
        for i in range(num_steps):
            j = i + 1
            out = decode(sd, tgt_in[:, :j], memory, tgt_mask[:j, :j], tgt_query=pos_queries[:, i:j],
                         tgt_query_mask=query_mask[i:j, :j], heads=dec_heads)
            p_i = head(out)
            logits.append(p_i)
            if j < num_steps:
                tgt_in[:, j] = p_i.squeeze(1).argmax(-1)
                if testing and (tgt_in == tok.eos_id).any(dim=-1).all():
                    break
        logits = torch.cat(logits, dim=1)
    else:
        tgt_in = torch.full((bs, 1), tok.bos_id, dtype=torch.long)
        logits = head(decode(sd, tgt_in, memory, tgt_query=pos_queries, heads=dec_heads))
    if refine_iters:
        query_mask = query_mask.clone()
        query_mask[torch.triu(torch.ones(num_steps, num_steps, dtype=torch.bool), 2)] = 0
        bos = torch.full((bs, 1), tok.bos_id, dtype=torch.long)
        for _ in range(refine_iters):
            tgt_in = torch.cat([bos, logits[:, :-1].argmax(-1)], dim=1)
            kpm = (tgt_in == tok.eos_id).int().cumsum(-1) > 0
            L = tgt_in.shape[1]
            out = decode(sd, tgt_in, memory, tgt_mask[:L, :L], kpm, tgt_query=pos_queries[:, :L],
                         tgt_query_mask=query_mask[:L, :L], heads=dec_heads)
            logits = head(out)
    return logits


# ---------------------------------------------------------------------------------------------- predictor
def predictor_transform(crops: Sequence[torch.Tensor], img_size=(32, 128)) -> torch.Tensor:
    """predictors/model.py:14-17,29: Resize(img_size, BICUBIC, antialias=True) + Normalize(0.5, 0.5) per crop"""
    out = [F.interpolate(t[None].float(), size=img_size, mode="bicubic", antialias=True, align_corners=False) for t in crops]
    return (torch.cat(out) - 0.5) / 0.5


def predictor_forward(sd, crops) -> torch.Tensor:
    return parseq_forward(sd, predictor_transform(crops))


def img2txt(sd, crops) -> List[str]:
    return Tokenizer().decode(predictor_forward(sd, crops))[0]


def calc_loss(sd, crops, labels: Sequence[str]) -> torch.Tensor:
    """predictors/model.py:41-57: per-sample cross entropy over the label's characters, clamped at 1"""
    tok = Tokenizer()
    preds = predictor_forward(sd, crops)
    gt = tok.encode(labels)
    losses = []
    for pred, g in zip(preds, gt):
        eos = int((g == 0).nonzero()[0].item())
        losses.append(torch.clamp(F.cross_entropy(pred[:eos - 1], g[1:eos]), max=1.0)[None])
    return torch.cat(losses)
