"""OCR scorer: the CPU oracle (oracle/parseq.py) against fixtures produced by the reference's own
PARSeq / Decoder / Tokenizer classes (tests/golden/make_parseq_golden.py), and the host side of the product module
(state-dict names, tokenizer, loud failure without a GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import parseq as OP
from udifftext_amd import synth

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "parseq_golden.npz"))


def _sd():
    keys, shapes = list(G["state_dict_keys"]), [eval(s) for s in G["state_dict_shapes"]]
    return {k: synth.synthetic_tensor("parseq." + k, sh) for k, sh in zip(keys, shapes)}


@pytest.fixture(scope="module")
def sd():
    return _sd()


def _close(a, b, tol):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape
    err = (a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()
    assert err < tol, f"rel rms {err:.3e} > {tol}"


def test_tokenizer_matches_reference():
    tok = OP.Tokenizer()
    assert [tok.eos_id, tok.bos_id, tok.pad_id, len(tok)] == list(G["ids"])
    np.testing.assert_array_equal(tok.encode(["Hello", "MI355X!", "a", ""]).numpy(), G["tok_encode"])
    labels, confs = tok.decode(torch.from_numpy(G["logits"]).softmax(-1))
    assert labels == list(G["tok_decode_labels"])
    np.testing.assert_allclose([c.prod().item() for c in confs], G["tok_decode_conf"], rtol=1e-4)


def test_encoder_restatement_matches_the_generators_vit(sd):
    # (not a reference pin: the fixture's memory comes from the stand-in ViT of the generator script — timm is absent)
    _close(OP.vit_encode(sd, torch.from_numpy(G["images"])), G["memory_from_stand_in_vit"], 2e-5)


def test_decode_teacher_forced_matches_reference(sd):
    mem = torch.from_numpy(G["memory_from_stand_in_vit"])
    tok = OP.Tokenizer()
    tgt = torch.from_numpy(G["tf_tgt"])
    L = tgt.shape[1]
    mask = torch.triu(torch.full((L, L), float("-inf")), 1)
    kpm = (tgt == tok.pad_id) | (tgt == tok.eos_id)
    _close(OP.decode(sd, tgt, mem, mask, kpm, tgt_query_mask=mask), G["tf_out"], 2e-5)
    cm, qm = torch.from_numpy(G["perm_content_mask"]), torch.from_numpy(G["perm_query_mask"])
    _close(OP.decode(sd, tgt[:, :-1], mem, cm, kpm[:, :-1], tgt_query_mask=qm), G["perm_out"], 2e-5)


def test_full_inference_matches_reference(sd):
    mem = torch.from_numpy(G["memory_from_stand_in_vit"])
    img = torch.from_numpy(G["images"])
    _close(OP.parseq_forward(sd, img, memory=mem), G["logits"], 5e-5)                    # AR + refinement
    _close(OP.parseq_forward(sd, img, max_length=7, memory=mem), G["logits_max7"], 5e-5)
    _close(OP.parseq_forward(sd, img, memory=mem, decode_ar=False), G["logits_nar"], 5e-5)
    _close(OP.parseq_forward(sd, img), G["logits"], 2e-4)                                # through the restated encoder


def test_predictor_module_names_and_loud_failure(sd):
    from sgm.modules.predictors.model import ParseqPredictor
    m = ParseqPredictor(ckpt_path=None)
    mine = {k: tuple(v.shape) for k, v in m.parseq.state_dict().items()}
    ref = {k: eval(s) for k, s in zip(G["state_dict_keys"], G["state_dict_shapes"])}
    assert mine == ref                                   # names and shapes of the reference checkpoint
    assert m.parseq.tokenizer.encode(["Hello", "MI355X!", "a", ""]).tolist() == G["tok_encode"].tolist()
    with pytest.raises(Exception):
        m([torch.rand(3, 40, 100)])        # CPU tensors: the HIP path refuses, there is no CPU fallback


def test_calc_loss_and_transform_shapes(sd):
    g = torch.Generator().manual_seed(3)
    crops = [torch.rand((3, 40, 100), generator=g), torch.rand((3, 25, 90), generator=g)]
    x = OP.predictor_transform(crops)
    assert x.shape == (2, 3, 32, 128) and float(x.abs().max()) <= 1.5          # (bicubic overshoot of a [0,1] image)
    loss = OP.calc_loss(sd, crops, ["ab", "MI3"])
    assert loss.shape == (2,) and bool((loss <= 1.0).all()) and bool((loss > 0).all())
