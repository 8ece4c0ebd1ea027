"""The torch-fp32 T5 / NCI oracle pinned against the reference's own outputs (goldens g1, g2, g3
captured by tools/capture_goldens.py from the vendored MEVI transformers fork).
Tolerances are f32 rounding of a different summation order: |diff| <= 2e-5 on O(1) activations."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from oracle import t5 as ot5

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_relative_position_buckets():
    g = np.load(os.path.join(GOLD, "g3_relative_buckets.npz"))
    for name, (ql, kl, bidir) in dict(enc32=(32, 32, True), dec6=(6, 6, False), enc200=(200, 200, True),
                                      dec150=(150, 150, False)).items():
        rel = np.arange(kl)[None, :] - np.arange(ql)[:, None]
        assert np.array_equal(ot5.relative_position_bucket(rel, bidir), g[name]), name


def test_tower_matches_reference_t5model():
    g = np.load(os.path.join(GOLD, "g2_t5_tower.npz"))
    cfg = json.loads(str(g["cfg"]))
    W = ot5.load_weights(g)
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    enc, hs = ot5.encoder(W, cfg, ids, mask, return_all=True)
    for i, h in enumerate(hs):
        assert np.abs(h.numpy() - g[f"enc_h{i}"]).max() <= 2e-5, i
    reps = ot5.tower_encode(W, cfg, ids, mask)
    assert np.abs(reps.numpy() - g["reps"]).max() <= 2e-5


def test_tower_matches_reference_t5model_at_passage_shape():
    """128-token passages (gen_doc_embedding, generate.py:116-187): ragged lengths incl. a full window and a
    3-token one."""
    g = np.load(os.path.join(GOLD, "g2p_t5_passage.npz"))
    cfg = json.loads(str(g["cfg"]))
    W = ot5.load_weights(g)
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    enc = ot5.encoder(W, cfg, ids, mask)
    assert np.abs(enc.numpy() - g["enc_last"]).max() <= 2e-5
    reps = ot5.tower_encode(W, cfg, ids, mask)
    assert np.abs(reps.numpy() - g["reps"]).max() <= 2e-5


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "g1_nci_*.npz"))))
def test_nci_generate_matches_reference(path):
    g = np.load(path)
    cfg = json.loads(str(g["cfg"]))
    W = ot5.load_weights(g)
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    dec, sc, enc, steps = ot5.nci_generate(W, cfg, ids, mask, cfg["beams"], return_steps=True)
    assert np.abs(enc.numpy() - g["enc_hidden"]).max() <= 2e-5
    assert np.array_equal(dec.numpy(), g["decoded"])                    # identical token matrices
    assert np.abs(sc.numpy() - g["scores"]).max() <= 5e-6               # hypothesis scores
    # step-0 logits of the reference (all beams identical at step 0): valid columns only
    V = g["step0_logits"].shape[1]
    B, R = ids.shape[0], cfg["beams"]
    for b in range(B):
        mine = next(l for (bb, p, l) in steps if bb == b and p == 0)[0].numpy()
        ref = g["step0_logits"][b * R]
        assert np.abs(mine - ref).max() <= 2e-4 * max(1.0, np.abs(ref[ref > -1e8]).max())
    codes = ot5.decode_token(dec, cfg["K"])
    assert codes.min() >= 0 and codes.max() < cfg["K"] and codes.shape[1] == cfg["M"]


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "g1t_nci_tree_*.npz"))))
def test_nci_generate_under_a_generic_prefix_tree_matches_reference(path):
    """Golden G1T: model.generate(decode_tree=TreeBuilder(share_sons=False) trie of `paths`) of the imported reference --
    dense trie, sparse tries (fewer live candidates than beams at some levels) and a one-path trie, whose result shows the
    reference's -1e9-seeded beams (the path R times, scores s, then -1e9 / (M + 1)^0.8)."""
    g = np.load(path)
    cfg = json.loads(str(g["cfg"]))
    W = ot5.load_weights(g)
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    dec, sc, _ = ot5.nci_generate_tree(W, cfg, ids, mask, cfg["beams"], g["paths"])
    assert np.array_equal(dec.numpy(), g["decoded"])
    ref = g["scores"]
    assert np.abs(sc.numpy() - ref).max() <= 5e-6 * np.maximum(1.0, np.abs(ref)).max()
    allowed = {tuple(int(c) for c in pth) for pth in g["paths"]}
    codes = ot5.decode_token(dec, cfg["K"]).numpy()
    assert all(tuple(int(c) for c in row) in allowed for row in codes)        # every hypothesis is a path of the trie


def test_bert_tower_oracle_matches_reference_bertmodel():
    """oracle/bert.py against the vendored BertModel's own outputs (mtype 'bert' towers: coCondenser / AR2)."""
    from oracle import bert as obert

    g = np.load(os.path.join(GOLD, "g8_bert_tower.npz"))
    cfg = json.loads(str(g["cfg"]))
    W = obert.load_weights(g)
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    hid = obert.encoder(W, cfg, ids, mask)
    valid = g["attention_mask"].astype(bool)
    assert np.abs(hid.numpy() - g["hidden"])[valid].max() <= 2e-5      # padded positions are not defined outputs
    assert np.abs(obert.tower_encode(W, cfg, ids, mask).numpy() - g["reps"]).max() <= 2e-5


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "g1a_nci_all_*.npz"))))
def test_nci_generate_all_matches_reference(path):
    """_generate_all (the use_topic_model ablation): the oracle's scores of all K**M code paths against the reference's
    generate(..., eval_all_documents=True) golden."""
    g = np.load(path)
    cfg = json.loads(str(g["cfg"]))
    W = ot5.load_weights(g)
    sc, _ = ot5.nci_generate_all(W, cfg, torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"]))
    assert sc.shape == g["all_scores"].shape and np.abs(sc.numpy() - g["all_scores"]).max() <= 5e-6
    # consistency with the beam search: its best hypothesis is the best of all paths
    dec, bsc, _ = ot5.nci_generate(W, cfg, torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"]), cfg["K"])
    R = cfg["K"]
    for b in range(len(g["input_ids"])):
        codes = ot5.decode_token(dec[b * R:b * R + 1], cfg["K"])[0]
        idx = int(sum(int(c) * cfg["K"] ** (cfg["M"] - 1 - p) for p, c in enumerate(codes)))
        assert abs(float(bsc[b * R]) - float(sc[b, idx])) <= 2e-6
