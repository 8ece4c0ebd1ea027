"""GPU parity of the T5 stacks and the NCI beam search (HIP kernels through the C ABI) against the
reference's own outputs (goldens g1/g2) and the torch-fp32 oracle.
Tolerance: hidden states |diff| <= 5e-5 on O(1) activations (summation order of f32 sums);
beam outputs: identical token matrices, hypothesis scores within 1e-5 (north star: identical
ranked lists / MRR within 1e-4)."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from mevi_amd import nci, t5
from oracle import t5 as ot5

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_bucket_tables_match_reference():
    g = np.load(os.path.join(GOLD, "g3_relative_buckets.npz"))
    for name, (ql, kl, bidir) in dict(enc32=(32, 32, True), dec6=(6, 6, False), enc200=(200, 200, True),
                                      dec150=(150, 150, False)).items():
        rel = np.arange(kl)[None, :] - np.arange(ql)[:, None]
        assert np.array_equal(t5.relative_position_bucket(rel, bidir), g[name]), name


def test_twin_tower_matches_reference_golden(cuda):
    g = np.load(os.path.join(GOLD, "g2_t5_tower.npz"))
    cfg = json.loads(str(g["cfg"]))
    tower = t5.TwinTower(nci.load_npz_weights(g), device=cuda, **cfg)
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    enc = tower.encoder.forward(tower.shared, ids.to(cuda), mask.to(cuda), pack=False)     # reference layout, all positions
    assert np.abs(enc.cpu().numpy() - g["enc_last"]).max() <= 5e-5
    packed = tower.encoder.forward(tower.shared, ids.to(cuda), mask.to(cuda), pack=True)  # real tokens only
    valid = g["attention_mask"].astype(bool)
    assert np.array_equal(packed.cpu().numpy()[valid], enc.cpu().numpy()[valid])           # bit-identical where it matters
    assert np.abs(packed.cpu().numpy()[~valid]).max() == 0 if (~valid).any() else True
    reps = tower.encode_query({"input_ids": ids, "attention_mask": mask})
    assert np.abs(reps.cpu().numpy() - g["reps"]).max() <= 5e-5
    # a mask with holes (no tokenizer produces one, the API allows it): the packed rows keep the padded layout for
    # attention, positions are the original ones -- same bits as the unpacked computation, end to end
    holes = mask.clone()
    holes[:, 2] = 0
    holes[0, 5:] = 0
    enc_h = tower.encoder.forward(tower.shared, ids.to(cuda), holes.to(cuda), pack=False)
    pk_h = tower.encoder.forward(tower.shared, ids.to(cuda), holes.to(cuda), pack=True)
    hv = holes.numpy().astype(bool)
    assert np.array_equal(pk_h.cpu().numpy()[hv], enc_h.cpu().numpy()[hv])
    tower.encoder.pack = False
    r0 = tower.encode_query({"input_ids": ids, "attention_mask": holes})
    tower.encoder.pack = True
    assert torch.equal(tower.encode_query({"input_ids": ids, "attention_mask": holes}), r0)


def test_passage_tower_matches_reference_golden(cuda):
    """The tied tower on 128-token passages (encode_passage, document_encoder.py:122-123) vs the reference's
    own T5Model outputs."""
    g = np.load(os.path.join(GOLD, "g2p_t5_passage.npz"))
    cfg = json.loads(str(g["cfg"]))
    tower = t5.TwinTower(nci.load_npz_weights(g), device=cuda, **cfg)
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    enc = tower.encoder.forward(tower.shared, ids.to(cuda), mask.to(cuda), pack=False)
    assert np.abs(enc.cpu().numpy() - g["enc_last"]).max() <= 5e-5
    packed = tower.encoder.forward(tower.shared, ids.to(cuda), mask.to(cuda), pack=True)
    valid = g["attention_mask"].astype(bool)
    assert np.array_equal(packed.cpu().numpy()[valid], enc.cpu().numpy()[valid])
    reps = tower.encode_passage({"input_ids": ids, "attention_mask": mask})
    assert np.abs(reps.cpu().numpy() - g["reps"]).max() <= 5e-5


def test_gen_doc_embedding_writes_the_reference_file_format(cuda, tmp_path):
    """generate.gen_doc_embedding: all_document_tokens.bin / all_document_masks.bin -> raw f32 [N, dim], batches
    that do not divide N, same values as one encode_passage call."""
    import generate

    g = np.load(os.path.join(GOLD, "g2p_t5_passage.npz"))
    cfg = json.loads(str(g["cfg"]))
    tower = t5.TwinTower(nci.load_npz_weights(g), device=cuda, **cfg)
    ids = np.concatenate([g["input_ids"], g["input_ids"][::-1]])[:11]
    mask = np.concatenate([g["attention_mask"], g["attention_mask"][::-1]])[:11]
    ids.astype(np.int64).tofile(tmp_path / "all_document_tokens.bin")
    mask.astype(np.int64).tofile(tmp_path / "all_document_masks.bin")
    out = str(tmp_path / "docemb.bin")
    generate.gen_doc_embedding(0, str(tmp_path), None, None, out, 4, cfg["d_model"], [0], 128, encoder=tower)
    got = np.fromfile(out, dtype=np.float32).reshape(-1, cfg["d_model"])
    want = tower.encode_passage({"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}).cpu().numpy()
    assert got.shape == (11, cfg["d_model"]) and np.array_equal(got, want)
    assert np.abs(got[:6] - g["reps"]).max() <= 5e-5
    assert sorted(os.listdir(tmp_path)) == ["all_document_masks.bin", "all_document_tokens.bin", "docemb.bin"]


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "g1_nci_*.npz"))))
def test_nci_generate_matches_reference_golden(cuda, path):
    g = np.load(path)
    cfg = json.loads(str(g["cfg"]))
    beams = cfg.pop("beams")
    model = nci.NCIModel(nci.load_npz_weights(g), device=cuda, **cfg)
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    dec, scores, enc, _ = model.generate(ids, mask, num_beams=beams, num_return_sequences=beams, max_length=cfg["M"] + 2)
    valid = g["attention_mask"].astype(bool)       # padded positions are don't-care outputs (0 in the packed encoder)
    assert np.abs(enc.cpu().numpy() - g["enc_hidden"])[valid].max() <= 5e-5
    assert np.array_equal(dec.cpu().numpy(), g["decoded"])
    assert np.abs(np.array(scores) - g["scores"]).max() <= 1e-5
    codes = nci.decode_token(dec, cfg["K"])
    assert np.array_equal(codes.cpu().numpy(), ot5.decode_token(torch.from_numpy(g["decoded"]), cfg["K"]).numpy())


def test_goldens_hold_with_the_layer_norms_folded_into_the_projections(cuda, monkeypatch):
    """MEVI_FOLD_NORM=1 (round 5; off by default because it measured slower, profiles/r05_fold_norm.txt): T5LayerNorm as a row scale
    behind the projection it feeds -- rmsnorm(x) W^T = rsqrt(mean x^2 + eps) (x (W (.) w_ln)^T) -- with the residual stream's image and
    block sums of squares written by the GEMM that produces it (mevi_gemm_nt_split_residual_stream / _normed_*).  The reference
    goldens hold at their tolerances, a row keeps its bits whether it travels alone (graph replay, latency kernels) or in a
    batch (tile stream), and the packed encoder equals the padded one."""
    from mevi_amd import ops

    monkeypatch.setattr(ops, "FOLD_NORM", True)
    g = np.load(os.path.join(GOLD, "g2_t5_tower.npz"))
    cfg = json.loads(str(g["cfg"]))
    tower = t5.TwinTower(nci.load_npz_weights(g), device=cuda, **cfg)
    assert tower.encoder.fold and tower.decoder.fold
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    enc = tower.encoder.forward(tower.shared, ids.to(cuda), mask.to(cuda), pack=False)
    assert np.abs(enc.cpu().numpy() - g["enc_last"]).max() <= 5e-5
    packed = tower.encoder.forward(tower.shared, ids.to(cuda), mask.to(cuda), pack=True)
    valid = g["attention_mask"].astype(bool)
    assert np.array_equal(packed.cpu().numpy()[valid], enc.cpu().numpy()[valid])
    reps = tower.encode_query({"input_ids": ids, "attention_mask": mask})
    assert np.abs(reps.cpu().numpy() - g["reps"]).max() <= 5e-5
    for a in (0, 3):                                                     # eager, capture, replay: one query alone
        for _ in range(3):
            one = tower.encode_query({"input_ids": ids[a:a + 1], "attention_mask": mask[a:a + 1]}, graph=True)
            assert torch.equal(one, reps[a:a + 1])
    for path in sorted(glob.glob(os.path.join(GOLD, "g1_nci_*.npz"))):
        g = np.load(path)
        cfg = json.loads(str(g["cfg"]))
        beams = cfg.pop("beams")
        model = nci.NCIModel(nci.load_npz_weights(g), device=cuda, **cfg)
        assert model.encoder.fold and model.decoder.fold
        ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
        dec, scores, enc, _ = model.generate(ids, mask, num_beams=beams, num_return_sequences=beams, max_length=cfg["M"] + 2)
        valid = g["attention_mask"].astype(bool)
        assert np.abs(enc.cpu().numpy() - g["enc_hidden"])[valid].max() <= 5e-5
        assert np.array_equal(dec.cpu().numpy(), g["decoded"])
        assert np.abs(np.array(scores) - g["scores"]).max() <= 1e-5
        for _ in range(3):
            d1, s1, _, _ = model.generate(ids[:1], mask[:1], num_beams=beams, graph=True)
            assert torch.equal(d1, dec[:beams]) and s1 == scores[:beams]


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "g1t_nci_tree_*.npz"))))
def test_nci_generate_under_a_generic_prefix_tree_matches_reference_golden(cuda, path):
    """Golden G1T: generate(decode_tree = the trie of the existing code paths, TreeBuilder(share_sons=False),
    MEVI/main_models.py:50-63 + generation_utils.py:803-818) -- CSR-children mode of the beam step (mevi_beam_step_tree_f32).
    Identical token matrices, scores within 1e-5 (relative for the reference's -1e9-seeded beams of the one-path trie);
    every hypothesis is a path of the trie; the prefix tables on / off and the graph replay change nothing."""
    g = np.load(path)
    cfg = json.loads(str(g["cfg"]))
    beams = cfg.pop("beams")
    model = nci.NCIModel(nci.load_npz_weights(g), device=cuda, **cfg)
    tree = nci.PrefixTree(g["paths"], cfg["M"], cfg["K"], cuda)
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    dec, scores, _, _ = model.generate(ids, mask, num_beams=beams, decode_tree=tree)
    assert np.array_equal(dec.cpu().numpy(), g["decoded"])
    ref = g["scores"]
    assert (np.abs(np.array(scores) - ref) <= 1e-5 * np.maximum(1.0, np.abs(ref))).all()
    allowed = {tuple(int(c) for c in pth) for pth in g["paths"]}
    codes = nci.decode_token(dec, cfg["K"]).cpu().numpy()
    assert all(tuple(int(c) for c in row) in allowed for row in codes)
    model.prefix_table_bytes, model._tables = 0, None                 # the adaptor evaluated per beam: same bits
    d2, s2, _, _ = model.generate(ids, mask, num_beams=beams, decode_tree=tree)
    assert torch.equal(d2, dec) and s2 == scores
    for _ in range(3):                                                # eager, capture, replay
        d3, s3, _, _ = model.generate(ids[:2], mask[:2], num_beams=beams, decode_tree=tree, graph=True)
        assert torch.equal(d3, dec[:2 * beams]) and s3 == scores[:2 * beams]
    # the trie of ALL K**M paths is the shared-sons tree: same beams as the default search (K >= R cases)
    if cfg["K"] >= beams and cfg["K"] ** cfg["M"] <= 1 << 20:
        full = np.stack(np.meshgrid(*[np.arange(cfg["K"])] * cfg["M"], indexing="ij"), -1).reshape(-1, cfg["M"])
        d4, s4, _, _ = model.generate(ids, mask, num_beams=beams, decode_tree=nci.PrefixTree(full, cfg["M"], cfg["K"], cuda))
        d5, s5, _, _ = model.generate(ids, mask, num_beams=beams)
        assert torch.equal(d4, d5) and s4 == s5


def _seeded_nci_weights(M, K, d, d_ff, heads, enc_layers=2, dec_layers=2, adaptor_layers=2):
    """Seeded random weights with the reference's initialiser scales (modeling_t5.py:603-633; nn.TransformerDecoder
    defaults for the adaptor; adaptor_embeddings ~ U(0,1) :1252) under the reference's state_dict names."""
    inner = heads * 64
    cfg = dict(M=M, K=K, d_model=d, d_ff=d_ff, num_heads=heads, d_kv=64, num_layers=enc_layers, num_decoder_layers=dec_layers,
               adaptor_layer_num=adaptor_layers, layer_norm_epsilon=1e-6, relative_attention_num_buckets=32)
    V = K * (M + 2) + 2
    W = {"shared.weight": torch.randn(1000, d), "decode_embeddings.weight": torch.randn(V, d),
         "adaptor_embeddings": torch.rand(1, 1, d), "adaptor_linear.weight": torch.randn(d * V, d) * d ** -0.5 * 0.3}
    W["lm_head.weight"] = W["decode_embeddings.weight"]
    for st, dec, layers in (("encoder", False, enc_layers), ("decoder", True, dec_layers)):
        for l in range(layers):
            p = f"{st}.block.{l}.layer"
            W[f"{p}.0.SelfAttention.q.weight"] = torch.randn(inner, d) * (d * 64) ** -0.5
            for n in "kv":
                W[f"{p}.0.SelfAttention.{n}.weight"] = torch.randn(inner, d) * d ** -0.5
            W[f"{p}.0.SelfAttention.o.weight"] = torch.randn(d, inner) * inner ** -0.5
            W[f"{p}.0.layer_norm.weight"] = 1 + 0.1 * torch.randn(d)
            ff = 1
            if dec:
                W[f"{p}.1.EncDecAttention.q.weight"] = torch.randn(inner, d) * (d * 64) ** -0.5
                for n in "kv":
                    W[f"{p}.1.EncDecAttention.{n}.weight"] = torch.randn(inner, d) * d ** -0.5
                W[f"{p}.1.EncDecAttention.o.weight"] = torch.randn(d, inner) * inner ** -0.5
                W[f"{p}.1.layer_norm.weight"] = 1 + 0.1 * torch.randn(d)
                ff = 2
            W[f"{p}.{ff}.DenseReluDense.wi.weight"] = torch.randn(d_ff, d) * d ** -0.5
            W[f"{p}.{ff}.DenseReluDense.wo.weight"] = torch.randn(d, d_ff) * d_ff ** -0.5
            W[f"{p}.{ff}.layer_norm.weight"] = 1 + 0.1 * torch.randn(d)
        W[f"{st}.block.0.layer.0.SelfAttention.relative_attention_bias.weight"] = torch.randn(32, heads) * 0.5
        W[f"{st}.final_layer_norm.weight"] = 1 + 0.1 * torch.randn(d)
    for l in range(adaptor_layers):
        p = f"adaptor.layers.{l}"
        for a in ("self_attn", "multihead_attn"):
            W[f"{p}.{a}.in_proj_weight"] = torch.randn(3 * d, d) * d ** -0.5
            W[f"{p}.{a}.in_proj_bias"] = torch.randn(3 * d) * 0.02
            W[f"{p}.{a}.out_proj.weight"] = torch.randn(d, d) * d ** -0.5
            W[f"{p}.{a}.out_proj.bias"] = torch.randn(d) * 0.02
        W[f"{p}.linear1.weight"], W[f"{p}.linear1.bias"] = torch.randn(2048, d) * d ** -0.5, torch.randn(2048) * 0.02
        W[f"{p}.linear2.weight"], W[f"{p}.linear2.bias"] = torch.randn(d, 2048) * 2048 ** -0.5, torch.randn(d) * 0.02
        for n in (1, 2, 3):
            W[f"{p}.norm{n}.weight"], W[f"{p}.norm{n}.bias"] = 1 + 0.1 * torch.randn(d), 0.05 * torch.randn(d)
    return W, cfg


@pytest.mark.parametrize("M,K,d,d_ff,heads,table_bytes,regime", [
    (4, 32, 768, 3072, 12, 6 << 30, "head matrices for every tabled position"),
    (4, 32, 768, 3072, 12, 6 << 30, "full depth: 12 encoder, 6 decoder, 4 adaptor layers (t5-base, marco_eval_nci_rq.sh)"),
    # BASELINE.json configs[2]: 3 levels of 256 codes.  Position 2 has 65 536 prefixes: its head matrices (17 GB at this
    # width) exceed the budget, so the tables hold ADAPTOR VECTORS ONLY there and the 257-column head GEMM runs per beam
    (3, 256, 256, 1024, 4, 6 << 30, "adaptor vectors only at position 2"),
    # budget too small for position 2 at all: the adaptor continues per beam from a cache assembled by lookup
    (3, 256, 256, 1024, 4, 200 << 20, "tables end before position 2"),
    # BASELINE.json configs[2] at its REAL width (t5-base: d 768, ff 3072, 12 x 64 heads), 2 + 2 + 2 layers
    (3, 256, 768, 3072, 12, 6 << 30, "t5-base width: adaptor vectors only at position 2"),
    # BASELINE.json configs[2] as it stands: (3, 256) at t5-base width AND full depth (pinned by the suite, not
    # only by bench.py's agreement sample)
    (3, 256, 768, 3072, 12, 6 << 30, "full depth: configs[2], adaptor vectors only at position 2"),
])
def test_base_shape_model_against_oracle(cuda, M, K, d, d_ff, heads, table_bytes, regime, record_property):
    """t5-base widths (d 768, ff 3072, 12x64 heads, adaptor heads of 96) with few layers, and the (3, 256) code shape
    of BASELINE.json at 64-wide heads, seeded random weights with the reference's initialiser scales: HIP path vs the
    torch-fp32 oracle, in each regime of the prefix tables.  The two FULL-DEPTH cases (BASELINE's own shapes) run 64 seeded
    queries (the oracle takes ~1.5 min of host time each) and report how many of the 640 beams sit in a
    near-tie swap: the count is asserted (<= 1 % of the beams) and written to gpurun_out/parity_certificate_*.json."""
    torch.manual_seed(0)
    R = 10
    full = regime.startswith("full depth")
    depth = dict(enc_layers=12, dec_layers=6, adaptor_layers=4) if full else {}
    W, cfg = _seeded_nci_weights(M, K, d, d_ff, heads, **depth)
    rng = np.random.default_rng(0)
    B, S = (64 if full else 5), 32
    ids = np.zeros((B, S), np.int64)
    mask = np.zeros((B, S), np.int64)
    for i in range(B):
        L = int(np.clip(rng.poisson(9) + 2, 3, S))
        ids[i, :L - 1] = rng.integers(3, 1000, size=L - 1)
        ids[i, L - 1] = 1
        mask[i, :L] = 1
    ids, mask = torch.from_numpy(ids), torch.from_numpy(mask)
    odec, osc, oenc = ot5.nci_generate(W, cfg, ids, mask, R)
    model = nci.NCIModel(W, device=cuda, prefix_table_bytes=table_bytes, **cfg)
    dec, sc, enc, _ = model.generate(ids, mask, num_beams=R)
    tab = model.tables()
    if "only at position 2" in regime:
        assert tab.levels == 3 and tab.tmat[1] is not None and tab.tmat[2] is None and tab.avec[2] is not None
    elif "end before" in regime:
        assert tab.levels == 2
    else:
        assert tab.levels == M and all(t is not None for t in tab.tmat)
    valid = mask.bool()                                            # padded positions: don't-care (0 when packed)
    assert (enc.cpu() - oenc)[valid].abs().max() <= 2e-4 * oenc.abs().max()
    sc = np.array(sc)
    assert np.abs(sc - osc.numpy()).max() <= 2e-4
    # identical beams wherever the oracle's neighbouring scores are separated by more than the tolerance
    # -- asserted per query, on the offending pair only: a row that differs from the oracle's at rank j must be the
    # oracle's row of a rank j' whose score is within the tolerance of rank j's (a swap inside a near-tie), nothing else
    got, want, oscq = dec.cpu().numpy().reshape(B, R, -1), odec.numpy().reshape(B, R, -1), osc.numpy().reshape(B, R)
    swapped, gaps = 0, []
    for i in range(B):
        for j in range(R):
            if (got[i, j] == want[i, j]).all():
                continue
            twins = [jj for jj in range(R) if (got[i, j] == want[i, jj]).all()]
            assert twins and abs(oscq[i, twins[0]] - oscq[i, j]) < 4e-4, (regime, i, j, twins, oscq[i].tolist())
            swapped += 1
            gaps.append(float(abs(oscq[i, twins[0]] - oscq[i, j])))
    cert = {"regime": regime, "M": M, "K": K, "queries": B, "beams": B * R, "beams_in_a_near_tie_swap": swapped,
            "swap_score_gaps": gaps, "beam_score_max_abs_diff": float(np.abs(sc - osc.numpy()).max()),
            "smallest_oracle_gap_between_neighbouring_beams": float(np.abs(np.diff(oscq, axis=1)).min())}
    record_property("parity_certificate", json.dumps(cert))
    assert swapped <= max(1, B * R // 100), cert                    # <= 1 % of the beams (and never more than a handful)
    if full:
        try:
            out = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "parity_certificate_%dx%d.json" % (M, K)), "w") as f:
                json.dump(cert, f, indent=1)
        except OSError:
            pass
    tower = t5.TwinTower(W, device=cuda, **{k: v for k, v in cfg.items() if k not in ("M", "K", "adaptor_layer_num")})
    reps = tower.encode_query({"input_ids": ids, "attention_mask": mask}).cpu()
    oreps = ot5.tower_encode(W, dict(cfg), ids, mask)
    assert (reps - oreps).abs().max() <= 2e-4 * oreps.abs().max()


def test_bert_tower_matches_reference_golden(cuda):
    """BERT-family tower (mtype 'bert', document_encoder.py:43-44) on the HIP kernels vs the vendored BertModel's
    outputs and the torch oracle."""
    from mevi_amd import bert
    from oracle import bert as obert

    g = np.load(os.path.join(GOLD, "g8_bert_tower.npz"))
    cfg = json.loads(str(g["cfg"]))
    W = nci.load_npz_weights(g)
    tower = bert.BertTower(W, cfg["num_hidden_layers"], cfg["num_attention_heads"], eps=cfg["layer_norm_eps"], device=cuda)
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    hid = tower.lm_q.forward(ids.to(cuda), mask.to(cuda), pack=False).cpu().numpy()
    valid = g["attention_mask"].astype(bool)
    assert np.abs(hid - g["hidden"])[valid].max() <= 5e-5
    packed = tower.lm_q.forward(ids.to(cuda), mask.to(cuda), pack=True).cpu().numpy()       # real tokens only
    assert np.array_equal(packed[valid], hid[valid]) and np.abs(packed[~valid]).max() == 0
    reps = tower.encode_query({"input_ids": ids, "attention_mask": mask}).cpu().numpy()
    assert np.abs(reps - g["reps"]).max() <= 5e-5
    assert np.array_equal(reps, tower.encode_passage({"input_ids": ids, "attention_mask": mask}).cpu().numpy())  # tied
    want = obert.tower_encode(obert.load_weights(g), cfg, ids, mask).numpy()
    assert np.abs(reps - want).max() <= 5e-5


def test_bert_towers_load_like_the_reference(cuda, tmp_path):
    """generate.load_document_encoder on the two BERT-family layouts the reference reads (generate.py:31-44):
    an HF directory (tied towers, 'bert.'-prefixed keys allowed) and an AR2 `.pkl` with ctx_model / question_model."""
    import generate

    g = np.load(os.path.join(GOLD, "g8_bert_tower.npz"))
    cfg = json.loads(str(g["cfg"]))
    sd = {k: v.clone() for k, v in nci.load_npz_weights(g).items()}
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    hf = tmp_path / "co-condenser-marco-retriever"
    hf.mkdir()
    (hf / "config.json").write_text(json.dumps(dict(model_type="bert", **cfg)))
    torch.save({"bert." + k: v for k, v in sd.items()}, hf / "pytorch_model.bin")
    enc = generate.load_document_encoder(str(hf), None, cuda)
    assert np.abs(enc.encode_query({"input_ids": ids, "attention_mask": mask}).cpu().numpy() - g["reps"]).max() <= 5e-5
    # AR2: separate question / context models; perturb the context model to see that encode_passage uses it
    ctx = {k: (v * 1.01 if k.endswith("query.weight") else v) for k, v in sd.items()}
    torch.save({"model_dict": {**{"question_model." + k: v for k, v in sd.items()},
                               **{"ctx_model." + k: v for k, v in ctx.items()}}}, tmp_path / "ar2g_marco_finetune.pkl")
    enc2 = generate.load_document_encoder(str(tmp_path / "ar2g_marco_finetune.pkl"), None, cuda)
    q = enc2.encode_query({"input_ids": ids, "attention_mask": mask}).cpu().numpy()
    p_ = enc2.encode_passage({"input_ids": ids, "attention_mask": mask}).cpu().numpy()
    assert np.abs(q - g["reps"]).max() <= 5e-5 and np.abs(p_ - q).max() > 1e-4


def test_results_do_not_depend_on_batch_grouping_or_stale_memory_at_base_shapes(cuda):
    """t5-base-shaped NCI model and tower (synthetic weights, tools/synth.py): beams, scores and embeddings are
    bit-identical whether the queries go through in one pass or in groups of 96 / 700 (GEMM tile shape, packed token
    count and attention grids all change), and with the allocator's free blocks poisoned with NaN (no kernel reads
    memory it did not write)."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth

    model, tower, _, _ = synth.build(cuda, 4, 32, 2048)
    nq = 1500
    ids, mask = synth.query_ids(nq, cuda, np.random.default_rng(7))

    def gen(b):
        parts = [model.generate(ids[a:a + b], mask[a:a + b], num_beams=10) for a in range(0, nq, b)]
        return torch.cat([p[0] for p in parts]), np.concatenate([np.asarray(p[1]) for p in parts])

    def emb(b):
        tower.batch_size = b
        return tower.encode_query({"input_ids": ids, "attention_mask": mask})

    d0, s0 = gen(nq)
    e0 = emb(2048)
    assert np.isfinite(s0).all() and bool(torch.isfinite(e0).all())
    for b in (96, 700):
        d, s = gen(b)
        assert torch.equal(d, d0) and np.array_equal(s, s0)
        assert torch.equal(emb(b), e0)
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 28,), float("nan"), device=cuda) for _ in range(8)]
    del junk
    d, s = gen(700)
    assert torch.equal(d, d0) and np.array_equal(s, s0) and torch.equal(emb(700), e0)
    # the per-prefix adaptor tables (PrefixTables) against the adaptor evaluated per beam per step: same bits
    tab = model.tables()
    # device-sized budget: head matrices for positions 0-3, and the final position's 32**4 adaptor vectors (3.2 GB, built in chunks)
    assert tab.levels == 5 and all(t is not None for t in tab.tmat[:4]) and tab.tmat[3].shape == (32 ** 3, 33 * 768)
    assert tab.tmat[4] is None and tab.avec[4].shape == (32 ** 4, 768)
    model.prefix_table_bytes = 0
    d, s = gen(700)
    assert torch.equal(d, d0) and np.array_equal(s, s0)
    model.prefix_table_bytes, model._tables = 6 << 30, None             # round 3's budget: the final position per beam (cache read in place)
    d, s = gen(nq)
    assert model.tables().levels == 4
    assert torch.equal(d, d0) and np.array_equal(s, s0)
    model.prefix_table_bytes, model._tables = 60 << 20, None           # tables for positions 0..2 only, adaptor vectors at 2
    d, s = gen(nq)
    tab = model.tables()
    assert tab.levels == 3 and tab.tmat[1] is not None and tab.tmat[2] is None and tab.avec[2] is not None
    assert torch.equal(d, d0) and np.array_equal(s, s0)


def test_device_sized_prefix_tables_for_the_3x256_codebook_keep_the_bits(cuda):
    """BASELINE.json configs[2] code shape at t5-base width: with no budget named the tables take half of the free device
    memory (at most 128 GiB, nci.default_table_bytes) -- room for the head matrices of all 65 536 two-code prefixes (52 GB),
    so position 2's 257-column head GEMM is a lookup, and for the adaptor outputs of the 16.7 M prefixes of the final position.  Same beams, same score bits as with the former 6 GB budget (adaptor
    vectors only at position 2, GEMM per beam)."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth

    free, _ = torch.cuda.mem_get_info(cuda)
    if free < (220 << 30):
        pytest.skip("needs a 288 GB device (103 GB of tables)")
    model, _, _, _ = synth.build(cuda, 3, 256, 512)
    del _
    ids, mask = synth.query_ids(300, cuda, np.random.default_rng(11))
    assert model.prefix_table_bytes is None
    d0, s0 = model.generate(ids, mask, num_beams=10)[:2]
    tab = model.tables()
    assert 0 < model.prefix_table_bytes <= (128 << 30)
    assert tab.levels == 4 and tab.tmat[2] is not None and tab.tmat[2].shape == (256 ** 2, 257 * 768)
    assert tab.tmat[3] is None and tab.avec[3].shape == (256 ** 3, 768)       # the final position: adaptor outputs, 51 GB
    model.prefix_table_bytes, model._tables = 6 << 30, None
    del tab
    torch.cuda.empty_cache()
    d1, s1 = model.generate(ids, mask, num_beams=10)[:2]
    tab = model.tables()
    assert tab.levels == 3 and tab.tmat[2] is None and tab.avec[2] is not None
    assert torch.equal(d0, d1) and np.array_equal(np.asarray(s0), np.asarray(s1))


def test_prefix_tables_sized_for_the_workload_cost_a_one_shot_eval_nothing(cuda):
    """a one-shot `main.py --mode eval` over MS MARCO dev (6980 queries) must not pay seconds of table build
    for prefixes it never visits.  With the run's query count passed (EvalRun.run -> NCIModel.expect_queries) a position is
    tabled only when queries x beams outnumber its prefixes: at (3, 256) that leaves positions 0 and 1 (the 65 536 two-code
    prefixes and the 16.7 M of the final position are not worth 69 800 beam rows) -- the first pass, build included, is not
    slower than with the former 6 GiB budget, which the device-sized default of a long-lived model loses to by seconds; same
    beams, same score bits under every policy.  (4, 32): positions 0..3 tabled, the 1 M prefixes of the final position not."""
    import sys
    import time

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth

    free, _ = torch.cuda.mem_get_info(cuda)
    if free < (220 << 30):
        pytest.skip("needs a 288 GB device (the long-lived policy tables 103 GB)")
    nq = 6980
    ids, mask = synth.query_ids(nq, cuda, np.random.default_rng(3))

    def first_pass(model):
        torch.cuda.synchronize()
        t = time.perf_counter()
        d, s = model.generate(ids, mask, num_beams=10)[:2]
        torch.cuda.synchronize()
        return (time.perf_counter() - t) * 1e3, d, np.asarray(s)

    model, _, _, _ = synth.build(cuda, 3, 256, None)
    del _
    model.prefix_table_bytes = 0
    model.generate(ids[:256], mask[:256], num_beams=10)               # kernels warm, allocator grown, no table yet
    model.prefix_table_bytes = None
    model.expect_queries(nq)
    t_work, d0, s0 = first_pass(model)
    tab = model.tables()
    assert tab.levels == 2 and tab.expected_queries == nq and tab.build_ms < 200, tab.describe()
    model.prefix_table_queries, model.prefix_table_bytes, model._tables = None, 6 << 30, None
    t_6g, d1, s1 = first_pass(model)
    assert model.tables().levels == 3
    assert torch.equal(d0, d1) and np.array_equal(s0, s1)
    assert t_work <= t_6g * 1.10, (t_work, t_6g)
    model.prefix_table_queries, model.prefix_table_bytes, model._tables = None, None, None
    t_all, d2, s2 = first_pass(model)                                   # the long-lived policy: every table that fits
    assert model.tables().levels == 4 and model.tables().build_ms > 500
    assert torch.equal(d0, d2) and np.array_equal(s0, s2)
    assert t_work < t_all, (t_work, t_all)
    # a larger workload announced later rebuilds (more positions pay); a smaller one keeps what exists
    model._tables = None
    model.expect_queries(1000)
    assert model.tables(10).levels == 2
    model.expect_queries(100000)
    assert model._tables is None and model.tables(10).levels == 3
    del model
    torch.cuda.empty_cache()
    model, _, _, _ = synth.build(cuda, 4, 32, None)
    del _
    model.expect_queries(nq)
    d0, s0 = model.generate(ids[:500], mask[:500], num_beams=10)[:2]
    assert model.tables().levels == 4 and model.tables().tmat[3] is not None
    model.prefix_table_queries, model._tables = None, None
    d1, s1 = model.generate(ids[:500], mask[:500], num_beams=10)[:2]
    assert model.tables().levels == 5
    assert torch.equal(d0, d1) and np.array_equal(np.asarray(s0), np.asarray(s1))


def test_graph_replay_of_small_batches_equals_the_eager_pass(cuda):
    """graph=True: batches of <= GRAPH_MAX_ROWS queries replay one captured HIP graph of the whole tower forward / beam
    search; first call eager, second captures, later ones replay with new inputs -- all bit-identical to the eager,
    packed pass of the same queries inside a large batch (which also pins the few-row GEMM kernel against the MFMA
    tiles end to end)."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth

    model, tower, _, _ = synth.build(cuda, 4, 32, None)
    ids, mask = synth.query_ids(40, cuda, np.random.default_rng(3))
    big_e = tower.encode_query({"input_ids": ids, "attention_mask": mask})
    big_d, big_s, _, _ = model.generate(ids, mask, num_beams=10)
    big_s = np.asarray(big_s).reshape(40, 10)
    for b in (1, 2, 24):                                 # (24: above the 8 rows graphs were limited to before round 6)
        for it, a in enumerate((0, 5, 11, 30) if b < 24 else (0, 5, 11, 16)):          # eager, capture, replay, replay
            q = {"input_ids": ids[a:a + b], "attention_mask": mask[a:a + b]}
            assert torch.equal(tower.encode_query(q, graph=True), big_e[a:a + b]), (b, it)
            d, s, enc, _ = model.generate(ids[a:a + b], mask[a:a + b], num_beams=10, graph=True)
            assert torch.equal(d, big_d[10 * a:10 * (a + b)]) and np.array_equal(np.asarray(s).reshape(b, 10), big_s[a:a + b])
    assert ("tower", 1, 32) in tower._graphs.graphs and any(k[0] == "generate" for k in model._graphs.graphs)
    # the default stays eager (packed) for a small batch
    assert torch.equal(tower.encode_query({"input_ids": ids[:2], "attention_mask": mask[:2]}), big_e[:2])
    assert len(tower._graphs.graphs) == 3


def test_captured_graphs_survive_a_larger_eager_pass(cuda):
    """the attention contexts' shared exponent array (ops._ctx_image) grows when a pass has more rows
    than it holds; a graph captured before holds the OLD array's address.  Capture the small-batch graphs, poison-proof the
    allocator with a pass of > 65536 context rows (which re-allocates the fill) plus scratch garbage over freed memory,
    replay: same bits as before."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth
    from mevi_amd import ops

    # (the fills are per process: an earlier test with a whole MS MARCO-sized pass has grown them -- start from fresh ones, the old
    # arrays retired as the product retires them)
    ops._EXP_FILL_RETIRED.extend(ops._EXP_FILL.values())
    ops._EXP_FILL.clear()
    model, tower, _, _ = synth.build(cuda, 4, 32, None)
    ids, mask = synth.query_ids(2400, cuda, np.random.default_rng(5))
    q = {"input_ids": ids[:2], "attention_mask": mask[:2]}
    for _ in range(2):                                   # eager, then capture
        e0 = tower.encode_query(q, graph=True)
        d0, s0, _, _ = model.generate(ids[:2], mask[:2], num_beams=10, graph=True)
    fills = {k: v.data_ptr() for k, v in ops._EXP_FILL.items()}
    assert fills and max(v.numel() for v in ops._EXP_FILL.values()) == 1 << 16
    full = {"input_ids": ids, "attention_mask": torch.ones_like(mask)}       # 2400 x 32 = 76800 context rows > 65536
    tower.encode_query(full)
    model.generate(ids, torch.ones_like(mask), num_beams=10)
    assert any(ops._EXP_FILL[k].data_ptr() != p for k, p in fills.items()), "the pass did not outgrow the fill"
    assert ops._EXP_FILL_RETIRED, "a superseded fill must be retired, not freed"
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 16,), 77, dtype=torch.int8, device=cuda) for _ in range(64)]   # lands on any freed 64 KB block
    torch.cuda.synchronize()
    e1 = tower.encode_query(q, graph=True)               # replay
    d1, s1, _, _ = model.generate(ids[:2], mask[:2], num_beams=10, graph=True)
    del junk
    assert torch.equal(e0, e1) and torch.equal(d0, d1) and np.array_equal(np.asarray(s0), np.asarray(s1))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "g1a_nci_all_*.npz"))))
def test_nci_generate_all_matches_reference_golden(cuda, path):
    """generate(..., eval_all_documents=True) = _generate_all (generation_utils.py:1013-1136, the use_topic_model
    ablation): scores of all K**M code paths vs the reference's golden; the depth-first block splitting (by query, by
    prefix) and the prefix tables change nothing."""
    g = np.load(path)
    cfg = json.loads(str(g["cfg"]))
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    outs = []
    for table_bytes, max_rows in ((6 << 30, 1 << 16), (0, 1 << 16), (6 << 30, 16), (0, 5), (1 << 14, 2 * cfg["K"])):
        model = nci.NCIModel(nci.load_npz_weights(g), device=cuda, prefix_table_bytes=table_bytes, **cfg)
        if max_rows == 1 << 16:
            dec, sc, enc, _ = model.generate(ids, mask, num_beams=1, num_return_sequences=1, eval_all_documents=True,
                                             max_length=cfg["M"] + 2)
            assert dec is None
        else:
            sc, enc = model.generate_all(ids, mask, max_rows=max_rows)
        assert np.abs(sc.cpu().numpy() - g["all_scores"]).max() <= 1e-5
        outs.append(sc)
    assert all(torch.equal(o, outs[0]) for o in outs[1:])          # row-wise operators: the blocking changes no bit
    # the beam search's best hypothesis is the best of all paths
    model = nci.NCIModel(nci.load_npz_weights(g), device=cuda, **cfg)
    dec, bsc, _, _ = model.generate(ids, mask, num_beams=cfg["K"])
    best = outs[0].max(1)
    assert np.abs(np.array(bsc).reshape(len(ids), -1)[:, 0] - best.values.cpu().numpy()).max() <= 2e-6
    codes = nci.decode_token(dec, cfg["K"]).view(len(ids), cfg["K"], cfg["M"])[:, 0]
    idx = sum(codes[:, p] * cfg["K"] ** (cfg["M"] - 1 - p) for p in range(cfg["M"]))
    assert torch.equal(idx.cpu(), best.indices.cpu())
