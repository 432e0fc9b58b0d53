"""Golden vectors for the OCR scorer, generated in the build container from the REAL reference
classes of /root/reference/src/parseq (strhub.models.parseq.system.PARSeq, modules.Decoder / DecoderLayer /
TokenEmbedding, strhub.data.utils.Tokenizer):

    python tests/golden/make_parseq_golden.py        -> tests/golden/parseq_golden.npz

What is and is not the reference here.  PARSeq's encoder is timm's VisionTransformer (timm~=0.6.5), which is not
installed in this image and not vendored by the reference.  To let the reference's own decoding code run, the
``timm`` import is satisfied by the stand-in ``_ViT`` below (the published ViT forward: patch conv, + pos_embed,
pre-LN blocks with eps 1e-6, final norm; parameter names as in timm's state dict).  Everything downstream of the
encoder output — the two-stream decoder, the autoregressive loop with its early exit, the refinement pass, the
masks, the head, the token embedding, the tokenizer — is the reference's code, unmodified.  The fixture therefore
pins the decoder side given ``memory``; the encoder stays "parity unpinned" (oracle/parseq.py header).
Weights: the name-keyed synthetic recipe (udifftext_amd/synth.py), as for the engine goldens.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src/parseq"
sys.path.insert(0, ROOT)
from udifftext_amd import synth  # noqa: E402


class _PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim, **kw):
        super().__init__()
        self.num_patches = (img_size[0] // patch_size[0]) * (img_size[1] // patch_size[1])
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=tuple(patch_size), stride=tuple(patch_size))

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class _Attn(nn.Module):
    def __init__(self, dim, heads, qkv_bias):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
        return self.proj(a.transpose(1, 2).reshape(B, N, C))


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(dim, hidden), nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class _Block(nn.Module):
    def __init__(self, dim, heads, mlp_ratio, qkv_bias):
        super().__init__()
        self.norm1, self.norm2 = nn.LayerNorm(dim, eps=1e-6), nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attn(dim, heads, qkv_bias)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class _ViT(nn.Module):
    """stand-in for timm.models.vision_transformer.VisionTransformer as modules.Encoder constructs it"""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.,
                 qkv_bias=True, drop_rate=0., attn_drop_rate=0., drop_path_rate=0., embed_layer=_PatchEmbed,
                 num_classes=0, global_pool='', class_token=False):
        super().__init__()
        assert num_classes == 0 and global_pool == '' and not class_token
        self.patch_embed = embed_layer(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches, embed_dim))
        self.blocks = nn.Sequential(*[_Block(embed_dim, num_heads, mlp_ratio, qkv_bias) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)

    def no_weight_decay(self):
        return {"pos_embed"}

    def forward_features(self, x):
        return self.norm(self.blocks(self.patch_embed(x) + self.pos_embed))


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _LM(nn.Module):
        _device = torch.device("cpu")

        def save_hyperparameters(self, *a, **k): pass
        def log(self, *a, **k): pass

    pl = mod("pytorch_lightning", LightningModule=_LM)
    pl.utilities = mod("pytorch_lightning.utilities")
    pl.utilities.types = mod("pytorch_lightning.utilities.types", STEP_OUTPUT=object)
    timm = mod("timm")
    timm.models = mod("timm.models")
    timm.models.vision_transformer = mod("timm.models.vision_transformer", VisionTransformer=_ViT, PatchEmbed=_PatchEmbed)

    def named_apply(fn, module, name="", depth_first=True, include_root=False):
        if not depth_first and include_root:
            fn(module=module, name=name)
        for cn, cm in module.named_children():
            named_apply(fn, cm, ".".join((name, cn)) if name else cn, depth_first, True)
        if depth_first and include_root:
            fn(module=module, name=name)
        return module

    timm.models.helpers = mod("timm.models.helpers", named_apply=named_apply)
    timm.optim = mod("timm.optim", create_optimizer_v2=lambda *a, **k: None)
    if "nltk" not in sys.modules:
        try:
            import nltk  # noqa: F401
        except Exception:
            mod("nltk", edit_distance=lambda a, b: 0)


def main():
    install_stubs()
    sys.path.insert(0, REF)
    from strhub.models.parseq.system import PARSeq
    from strhub.data.utils import Tokenizer

    charset = "0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~"
    torch.manual_seed(0)
    model = PARSeq(charset_train=charset, charset_test=charset, max_label_length=25, batch_size=1, lr=1e-3, warmup_pct=0.1,
                   weight_decay=0.0, img_size=(32, 128), patch_size=(4, 8), embed_dim=384, enc_num_heads=6, enc_mlp_ratio=4,
                   enc_depth=12, dec_num_heads=12, dec_mlp_ratio=4, dec_depth=1, perm_num=6, perm_forward=True,
                   perm_mirrored=True, decode_ar=True, refine_iters=1, dropout=0.1).eval()
    synth.fill_module_(model, prefix="parseq.")
    keys = list(model.state_dict().keys())
    out = {"state_dict_keys": np.array(keys), "state_dict_shapes": np.array([str(tuple(v.shape)) for v in model.state_dict().values()])}

    g = torch.Generator().manual_seed(5)
    images = torch.rand((3, 3, 32, 128), generator=g) * 2 - 1
    with torch.no_grad():
        memory = model.encode(images)
        out["images"] = images.numpy()
        out["memory_from_stand_in_vit"] = memory.numpy()
        # the reference's full inference path (AR decoding + 1 refinement pass), default and capped lengths
        out["logits"] = model(images).numpy()
        out["logits_max7"] = model(images, max_length=7).numpy()
        model.decode_ar = False
        out["logits_nar"] = model(images).numpy()
        model.decode_ar = True
        # one teacher-forced decode with masks and a padding mask (system.py:83-95)
        tok = Tokenizer(charset)
        tgt = tok.encode(["Hello", "MI355X!", "a"])
        L = tgt.shape[1]
        tgt_mask = torch.triu(torch.full((L, L), float("-inf")), 1)
        kpm = (tgt == tok.pad_id) | (tgt == tok.eos_id)
        out["tf_tgt"] = tgt.numpy()
        out["tf_out"] = model.decode(tgt, memory, tgt_mask, kpm, tgt_query_mask=tgt_mask).numpy()
        # a permuted-order mask pair as training builds them (system.py:176-190), teacher-forced through decode
        perm = torch.tensor([0, 3, 1, 4, 2, 5, 6, 7, 8][:L])
        cm, qm = model.generate_attn_masks(perm)
        out["perm_content_mask"], out["perm_query_mask"] = cm.numpy(), qm.numpy()
        out["perm_out"] = model.decode(tgt[:, :-1], memory, cm, kpm[:, :-1], tgt_query_mask=qm).numpy()
        # tokenizer
        out["tok_encode"] = tok.encode(["Hello", "MI355X!", "a", ""]).numpy()
        probs = model(images).softmax(-1)
        labels, confs = tok.decode(probs)
        out["tok_decode_labels"] = np.array(labels)
        out["tok_decode_conf"] = np.array([c.prod().item() for c in confs])
        out["ids"] = np.array([tok.eos_id, tok.bos_id, tok.pad_id, len(tok)])
    np.savez_compressed(os.path.join(HERE, "parseq_golden.npz"), **out)
    print("[golden] parseq_golden.npz:", {k: getattr(v, "shape", None) for k, v in out.items()})
    print("decoded:", labels)


if __name__ == "__main__":
    main()
