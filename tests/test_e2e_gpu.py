"""End-to-end on the GPU: the eval driver (main.py's EvalRun) and the dense CLI chain on a miniature
of config C4, checked against a CPU restatement of infer() built from the oracle pieces, then fed to
the consumers exactly like marco_ensemble.sh does."""
import json
import os
import pickle
import shutil
import subprocess
import sys
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import dense as odense
from oracle import rq as orq
from oracle import t5 as ot5

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


class FakeTokenizer:
    """Deterministic stand-in for the SentencePiece tokenizer (tokenisation is the boundary)."""

    def __init__(self, vocab):
        self.vocab = vocab

    def batch_encode_plus(self, texts, max_length=32, padding="max_length", truncation=True, return_tensors="pt"):
        ids = np.zeros((len(texts), max_length), np.int64)
        mask = np.zeros((len(texts), max_length), np.int64)
        for i, t in enumerate(texts):
            toks = [3 + (hash_(w) % (self.vocab - 3)) for w in t.split()][: max_length - 1] + [1]
            ids[i, :len(toks)] = toks
            mask[i, :len(toks)] = 1
        return {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}


def hash_(w):
    h = 2166136261
    for ch in w.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


def _build_mini(d, nci_golden, M, bits, R, n_random=2000):
    """A miniature MS MARCO: NCI checkpoint from a reference golden, T5-ANCE-shaped tower, corpus built around the clusters
    the (random-weight) model actually emits, RQ codebook, query file."""
    g = np.load(os.path.join(GOLD, nci_golden))
    cfg = json.loads(str(g["cfg"]))
    cfg.pop("beams", None)
    W = ot5.load_weights(g)
    tw = np.load(os.path.join(GOLD, "g2_t5_tower.npz"))
    TW = ot5.load_weights(tw)
    tcfg = json.loads(str(tw["cfg"]))
    os.makedirs(d / "ckpts" / "t5-ance")
    os.makedirs(d / "origin")
    os.makedirs(d / "ance")
    torch.save({"state_dict": {"model." + k: v for k, v in W.items()}}, d / "ckpts" / "nci.ckpt")
    torch.save(TW, d / "ckpts" / "t5-ance" / "pytorch_model.bin")
    json.dump(dict(d_model=32, d_ff=64, num_heads=4, d_kv=8, num_layers=2, num_decoder_layers=2), open(d / "ckpts" / "t5-ance" / "config.json", "w"))
    rng = np.random.default_rng(0)
    dim, K = 32, 2 ** bits
    queries = [" ".join(f"w{rng.integers(0, 50)}" for _ in range(rng.integers(3, 12))) + f" q{i}" for i in range(23)]
    # corpus built around the clusters the (random-weight) model actually emits, so that the fine
    # stage has documents to rank: doc = sum_j C[j][code_j] + noise for beam code paths + random paths
    enc = FakeTokenizer(512).batch_encode_plus(queries)
    dec, _, _ = ot5.nci_generate(W, cfg, enc["input_ids"], enc["attention_mask"], R)
    beam_codes = ot5.decode_token(dec, K).numpy()
    C = (rng.standard_normal((M, K, dim)) * (1.0 / np.arange(1, M + 1))[:, None, None]).astype(np.float32)
    C[0] *= 3.0
    paths = np.concatenate([np.repeat(beam_codes[::3], 6, axis=0), rng.integers(0, K, size=(n_random, M))])
    rng.shuffle(paths)
    N = len(paths)
    emb = sum(C[j][paths[:, j]] for j in range(M)).astype(np.float32) + 0.01 * rng.standard_normal((N, dim)).astype(np.float32)
    emb.tofile(d / "ance" / "docemb.bin")
    torch.save(torch.nn.Parameter(torch.from_numpy(C)), d / "ance" / f"rqcodebook{M}_{bits}.pt")
    gts = [[int(x) for x in rng.choice(N, size=1 + i % 2, replace=False)] for i in range(len(queries))]
    with open(d / "origin" / "dev_mevi_dedup.tsv", "w") as f:
        for q, g_ in zip(queries, gts):
            f.write(f"{q}\t{','.join(map(str, g_))}\n")
    args = Namespace(subvector_num=M, subvector_bits=bits, num_return_sequences=R, adaptor_layer_num=2, model_info="base",
                     nci_ckpt=str(d / "ckpts" / "nci.ckpt"), ckpt_dir=str(d / "ckpts"), embedding_path=str(d / "ance" / "docemb.bin"),
                     pq_path=str(d / "ance" / f"rqcodebook{M}_{bits}.pt"), pq_cluster_path=str(d / "ance" / f"rqclus{M}_{bits}.pkl"),
                     custom_save_path=str(d / "ance" / "nci_result_rq45_top10.tsv"), save_hard_neg=N, length_penalty=0.8,
                     eval_batch_size=4, recall_num=[1, 5, 10, 20, 50, 100], metric_path=str(d / "logs" / "m.txt"),
                     data_dir=str(d / "origin"), n_test=-1)
    return dict(dir=d, args=args, W=W, cfg=cfg, TW=TW, tcfg=tcfg, emb=emb, C=C, queries=queries, gts=gts, N=N)


@pytest.fixture(scope="module")
def mini(tmp_path_factory):
    return _build_mini(tmp_path_factory.mktemp("marco"), "g1_nci_M4_K32_R10.npz", 4, 5, 10)


@pytest.fixture(scope="module")
def mini_small(tmp_path_factory):
    """3 levels of 4 codes (64 code paths): small enough for _generate_all, which scores EVERY path of every query."""
    return _build_mini(tmp_path_factory.mktemp("marco_small"), "g1a_nci_all_M3_K4.npz", 3, 2, 4, n_random=600)


def test_eval_driver_matches_cpu_restatement(cuda, mini):
    from mevi_amd.evalrun import EvalRun, load_queries

    a = mini["args"]
    tok = FakeTokenizer(512)
    run = EvalRun(a, tokenizer=tok, device=cuda)
    out = run.run(load_queries(a.data_dir))
    prefix = a.custom_save_path[:-4]
    coarse = [l.rstrip("\n").split("\t") for l in open(prefix + "_coarse.tsv")]
    fine = [l.rstrip("\n").split("\t") for l in open(prefix + "_fine.tsv")]
    hn = [l.rstrip("\n").split("\t") for l in open(f"{prefix}_hn{a.save_hard_neg}.tsv")]
    assert len(coarse) == len(fine) == len(hn) == len(mini["queries"])
    # ---- CPU restatement of infer() from the oracle pieces
    enc = tok.batch_encode_plus(mini["queries"])
    ids, mask = enc["input_ids"], enc["attention_mask"]
    dec, sc, _ = ot5.nci_generate(mini["W"], mini["cfg"], ids, mask, 10)
    codes = ot5.decode_token(dec, 32).view(len(ids), 10, 4).numpy()
    sc = sc.numpy().reshape(len(ids), 10)
    qemb = ot5.tower_encode(mini["TW"], mini["tcfg"], ids, mask).numpy()
    cluster, mapping = orq.cluster_dict(orq.rq_encode(mini["emb"], mini["C"]))
    assert pickle.load(open(a.pq_cluster_path, "rb")) == cluster          # GPU RQ encode wrote the reference's pickles
    assert pickle.load(open(a.pq_cluster_path.replace("clus", "mapping"), "rb")) == mapping
    nd = 0
    for i, q in enumerate(mini["queries"]):
        assert coarse[i][0] == fine[i][0] == hn[i][0] == q
        assert eval(coarse[i][1]) == codes[i].tolist()                      # identical beam clusters
        assert np.abs(np.array(eval(coarse[i][3])) - sc[i]).max() <= 1e-5
        assert eval(coarse[i][2]) == [list(mapping[g]) for g in mini["gts"][i]]
        docs = [d for c in codes[i].tolist() for d in cluster.get(tuple(c), [])]
        nd += len(docs)
        got_docs = eval(fine[i][1])
        assert sorted(got_docs) == sorted(docs) and eval(fine[i][2]) == mini["gts"][i]
        if docs:
            ref = mini["emb"][docs] @ qemb[i]
            got_s = np.array([float(x) for x in hn[i][3].split(",")])
            order = np.argsort(-ref, kind="stable")
            assert np.abs(got_s - ref[order]).max() <= 2e-4
            assert [int(x) for x in hn[i][2].split(",")] == got_docs
            gaps = np.abs(np.diff(ref[order]))
            firm = np.concatenate([[True], gaps > 1e-3]) & np.concatenate([gaps > 1e-3, [True]])
            assert all(got_docs[j] == docs[order[j]] for j in np.nonzero(firm)[0])
        gs = np.array([float(x) for x in hn[i][1].split(",")])
        assert np.abs(gs - mini["emb"][mini["gts"][i]] @ qemb[i]).max() <= 2e-4
    assert nd > 50, "fixture should populate beam clusters"
    assert abs(out["ndoc"] - nd / len(mini["queries"])) < 1e-9
    assert os.path.exists(a.metric_path) and "ndocs@cluster10" in open(a.metric_path).read()


def test_eval_driver_with_the_trie_of_populated_clusters(cuda, mini, tmp_path, monkeypatch):
    """MEVI_DECODE_TREE=clusters: the beams are held to the generic trie of the code paths that own a cluster
    (TreeBuilder(share_sons=False) over the mapping, MEVI/main_models.py:50-63,1707-1728): the coarse log equals the oracle's
    tree search over the same paths, every beam cluster is populated, and the fine list is the union of those clusters."""
    import copy

    from mevi_amd.evalrun import EvalRun, load_queries

    monkeypatch.setenv("MEVI_DECODE_TREE", "clusters")
    a = copy.copy(mini["args"])
    a.custom_save_path = str(tmp_path / "trie.tsv")
    a.metric_path = str(tmp_path / "trie_metrics.txt")       # the module's shared metric file is compared by later tests
    tok = FakeTokenizer(512)
    run = EvalRun(a, tokenizer=tok, device=cuda)
    run.run(load_queries(a.data_dir))
    prefix = a.custom_save_path[:-4]
    coarse = [l.rstrip("\n").split("\t") for l in open(prefix + "_coarse.tsv")]
    fine = [l.rstrip("\n").split("\t") for l in open(prefix + "_fine.tsv")]
    cluster, _ = orq.cluster_dict(orq.rq_encode(mini["emb"], mini["C"]))
    paths = np.array(sorted(cluster), dtype=np.int64)
    enc = tok.batch_encode_plus(mini["queries"])
    ids, mask = enc["input_ids"], enc["attention_mask"]
    dec, sc, _ = ot5.nci_generate_tree(mini["W"], mini["cfg"], ids, mask, 10, paths)
    codes = ot5.decode_token(dec, 32).view(len(ids), 10, 4).numpy()
    sc = sc.numpy().reshape(len(ids), 10)
    changed = 0
    for i in range(len(mini["queries"])):
        got = eval(coarse[i][1])
        assert got == codes[i].tolist() and np.abs(np.array(eval(coarse[i][3])) - sc[i]).max() <= 1e-5
        assert all(tuple(c) in cluster for c in got)                          # no beam spent on an empty cluster
        assert sorted(eval(fine[i][1])) == sorted(d for c in got for d in cluster[tuple(c)])
    base = [l.rstrip("\n").split("\t") for l in open(mini["args"].custom_save_path[:-4] + "_coarse.tsv")] \
        if os.path.exists(mini["args"].custom_save_path[:-4] + "_coarse.tsv") else None
    if base is not None:                                                      # the shared-sons run of the first test
        changed = sum(eval(b[1]) != eval(c[1]) for b, c in zip(base, coarse))
        assert changed > 0                                                    # the trie did change some beam lists


def test_recall_levels_coarse_and_fine(cuda, mini, tmp_path):
    """--recall_level coarse | fine (main_models.py:3736,3781,4103-4201): the same beams and fine lists as 'both', each with
    its own result tuples, log files and metric keys -- cluster ranks at the cut-offs <= R and no tower pass for 'coarse',
    fine ranks + the found-at-all `cluster<R>` key for 'fine'."""
    import main
    from mevi_amd.evalrun import EvalRun, load_queries, summarize

    outs, files = {}, {}
    for level in ("both", "coarse", "fine"):
        a = Namespace(**vars(mini["args"]))
        a.recall_level = level
        a.recall_num = [1, 5, 10, 20, 50, 100] if level != "coarse" else [1, 5, 10]           # main.py:750-752
        a.custom_save_path, a.metric_path = str(tmp_path / level / "out.tsv"), str(tmp_path / level / "m.txt")
        os.makedirs(tmp_path / level)
        run = EvalRun(a, tokenizer=FakeTokenizer(512), device=cuda)
        if level == "coarse":
            run.tower = None                                                                  # must not be needed
        outs[level] = run.run(load_queries(a.data_dir))
        files[level] = {f: open(tmp_path / level / f, "rb").read() for f in sorted(os.listdir(tmp_path / level))}
    R = mini["args"].num_return_sequences
    assert sorted(files["coarse"]) == ["m.txt", "out_coarse.tsv", f"out_hn{mini['args'].save_hard_neg}.tsv"]
    assert files["coarse"]["out_coarse.tsv"] == files["both"]["out_coarse.tsv"] and files["coarse"][f"out_hn{mini['args'].save_hard_neg}.tsv"] == b""
    assert sorted(files["fine"]) == ["m.txt", "out_fine.tsv", f"out_hn{mini['args'].save_hard_neg}.tsv"]
    assert files["fine"]["out_fine.tsv"] == files["both"]["out_fine.tsv"]
    assert files["fine"][f"out_hn{mini['args'].save_hard_neg}.tsv"] == files["both"][f"out_hn{mini['args'].save_hard_neg}.tsv"]
    both, coarse, fine = outs["both"], outs["coarse"], outs["fine"]
    assert coarse["recall"] == {k: both["cluster_recall"][k] for k in (1, 5, 10)} and "cluster_recall" not in coarse
    assert coarse["mrr"] == {k: both["cluster_mrr"][k] for k in (1, 5, 10)} and coarse["ndoc"] == both["ndoc"]
    assert {k: v for k, v in fine["recall"].items() if k != f"cluster{R}"} == both["recall"] and "cluster_recall" not in fine
    assert fine["recall"][f"cluster{R}"] >= fine["recall"][100] and fine["ndoc"] == both["ndoc"]
    assert [l.split()[0] for l in files["coarse"]["m.txt"].decode().splitlines()] == \
        [f"{n}{k}" for n in ("recall", "mrr", "hitrate") for k in (1, 5, 10)] + [f"ndocs@cluster{R}:"]
    # the command line: the reference's own filter of recall_num, every level accepted
    argv = ["--mode", "eval", "--data_dir", "x", "--codebook", "1", "--pq_type", "rq", "--query_encoder", "twin",
            "--document_encoder", "ance", "--pq_path", "p", "--pq_cluster_path", "c", "--embedding_path", "e",
            "--custom_save_path", "o.tsv", "--nci_ckpt", "n", "--num_return_sequences", "10", "--subvector_num", "4",
            "--subvector_bits", "5"]
    assert main.parsers_parser(argv + ["--recall_level", "coarse"]).recall_num == [1, 5, 10]
    for level in ("both", "coarse", "fine"):
        main.check_supported(main.parsers_parser(argv + ["--recall_level", level]))
    with pytest.raises(SystemExit):
        main.check_supported(main.parsers_parser(argv + ["--recall_level", "finesampleloss"]))


def test_eval_outputs_do_not_depend_on_the_device_batch(cuda, mini, tmp_path):
    """marco_eval_nci_rq.sh passes --eval_batch_size 2; the driver feeds the GPU --device_batch_size queries per pass.
    Every log file must be byte-identical whatever the grouping (here 1, 4 and all 23 queries per pass)."""
    from mevi_amd.evalrun import EvalRun, load_queries

    blobs = []
    for j, (ebs, dbs) in enumerate([(1, None), (4, None), (2, 512)]):
        a = Namespace(**vars(mini["args"]))
        a.eval_batch_size, a.device_batch_size = ebs, dbs
        a.custom_save_path = str(tmp_path / f"r{j}" / "out.tsv")
        a.metric_path = str(tmp_path / f"r{j}" / "m.txt")
        os.makedirs(tmp_path / f"r{j}")
        EvalRun(a, tokenizer=FakeTokenizer(512), device=cuda).run(load_queries(a.data_dir))
        prefix = a.custom_save_path[:-4]
        blobs.append([open(p, "rb").read() for p in (prefix + "_coarse.tsv", prefix + "_fine.tsv",
                                                     f"{prefix}_hn{a.save_hard_neg}.tsv", a.metric_path)])
    assert blobs[0] == blobs[1] == blobs[2] and all(len(b) > 0 for b in blobs[0])
    # --knn_topk_by_step 1 outside the brute-force mode (main_models.py:3919-3995: running top-pool over the cluster
    # chunks): the fine lists are the pool_size best of the full lists, the candidate counts stay
    a = Namespace(**vars(mini["args"]))
    a.knn_topk_by_step, a.recall_num = 1, [1, 3]
    a.custom_save_path, a.metric_path = str(tmp_path / "topk" / "out.tsv"), str(tmp_path / "topk" / "m.txt")
    os.makedirs(tmp_path / "topk")
    EvalRun(a, tokenizer=FakeTokenizer(512), device=cuda).run(load_queries(a.data_dir))
    full = [l.split("\t") for l in blobs[0][1].decode().splitlines()]
    cut = [l.split("\t") for l in open(a.custom_save_path[:-4] + "_fine.tsv").read().splitlines()]
    assert len(full) == len(cut) and any(len(eval(f[1])) > 3 for f in full)
    for f, c in zip(full, cut):
        assert c[0] == f[0] and eval(c[1]) == eval(f[1])[:3] and c[2:] == f[2:]


def test_query_embeddings_of_generate_py_can_replace_the_second_tower_pass(cuda, mini, tmp_path):
    """main.py --query_embedding_path: the fine stage reads the file generate.py wrote; every output byte-identical to
    the run that encodes the queries again (the reference's behaviour)."""
    import generate
    from mevi_amd.evalrun import EvalRun, load_queries, load_tower_weights
    from mevi_amd.t5 import TwinTower

    a0 = mini["args"]
    tok = FakeTokenizer(512)
    tw, dims = load_tower_weights(os.path.join(a0.ckpt_dir, "t5-ance"))
    qpath = str(tmp_path / "query_emb.bin")
    generate.gen_query_embedding(0, os.path.join(a0.data_dir, "dev_mevi_dedup.tsv"), None, None, None, qpath, 128, 32, [0],
                                 tokenizer=tok, encoder=TwinTower(tw, dims=dims, device=cuda))
    blobs = []
    for j, qp in enumerate([None, qpath]):
        a = Namespace(**vars(a0))
        a.query_embedding_path = qp
        a.custom_save_path, a.metric_path = str(tmp_path / f"r{j}" / "out.tsv"), str(tmp_path / f"r{j}" / "m.txt")
        os.makedirs(tmp_path / f"r{j}")
        run = EvalRun(a, tokenizer=tok, device=cuda)
        if qp:
            run.tower = None                      # must not be needed
        run.run(load_queries(a.data_dir))
        prefix = a.custom_save_path[:-4]
        blobs.append([open(p, "rb").read() for p in (prefix + "_coarse.tsv", prefix + "_fine.tsv",
                                                     f"{prefix}_hn{a.save_hard_neg}.tsv", a.metric_path)])
    assert blobs[0] == blobs[1] and all(len(b) > 0 for b in blobs[0])


def test_whole_model_checkpoint_and_bad_checkpoints(cuda, mini, tmp_path):
    """--infer_ckpt (try_load_ckpt's whole-model branch, MEVI/main.py:198-230): NCI weights under `model.`, the query
    tower under `document_encoder.lm_q.`, the RQ codebook as `pq.codebook` (pq.initialize is skipped), the filtered
    relative_attention_bias keys ignored -- every output byte-identical to the --nci_ckpt run.  A checkpoint whose
    tensors do not fit the model is refused with the reference's `Bad parameter` report instead of being copied blindly."""
    from mevi_amd.evalrun import EvalRun, load_queries

    a0 = mini["args"]
    tok = FakeTokenizer(512)
    whole = {"model." + k: v for k, v in mini["W"].items()}
    whole.update({"document_encoder.lm_q." + k: v for k, v in mini["TW"].items()})
    whole.update({"document_encoder.lm_p." + k: v for k, v in mini["TW"].items()})
    whole["pq.codebook"] = torch.from_numpy(mini["C"])
    whole["model.decoder.block.0.layer.1.EncDecAttention.relative_attention_bias.weight"] = torch.full((32, 4), float("nan"))
    torch.save({"state_dict": whole}, tmp_path / "whole.ckpt")
    # the tower directory and the codebook file hold OTHER values: the run must take them from the checkpoint
    os.makedirs(tmp_path / "ckpts" / "t5-ance")
    torch.save({k: v + 1.0 for k, v in mini["TW"].items()}, tmp_path / "ckpts" / "t5-ance" / "pytorch_model.bin")
    shutil.copy(os.path.join(a0.ckpt_dir, "t5-ance", "config.json"), tmp_path / "ckpts" / "t5-ance" / "config.json")
    blobs = []
    for j in range(2):
        a = Namespace(**vars(a0))
        a.custom_save_path, a.metric_path = str(tmp_path / f"r{j}" / "out.tsv"), str(tmp_path / f"r{j}" / "m.txt")
        os.makedirs(tmp_path / f"r{j}")
        if j == 1:
            a.nci_ckpt, a.infer_ckpt, a.ckpt_dir, a.pq_path = None, str(tmp_path / "whole.ckpt"), str(tmp_path / "ckpts"), "/nonexistent.pt"
        EvalRun(a, tokenizer=tok, device=cuda).run(load_queries(a.data_dir))
        prefix = a.custom_save_path[:-4]
        blobs.append([open(p, "rb").read() for p in (prefix + "_coarse.tsv", prefix + "_fine.tsv",
                                                     f"{prefix}_hn{a.save_hard_neg}.tsv", a.metric_path)])
    assert blobs[0] == blobs[1] and all(len(b) > 0 for b in blobs[0])
    # a checkpoint trained with another codebook size (decode vocabulary of K = 16 instead of 32)
    bad = {"model." + k: v for k, v in mini["W"].items()}
    bad["model.decode_embeddings.weight"] = bad["model.decode_embeddings.weight"][:16 * 6 + 2].clone()
    torch.save({"state_dict": bad}, tmp_path / "bad.ckpt")
    a = Namespace(**vars(a0))
    a.nci_ckpt = str(tmp_path / "bad.ckpt")
    with pytest.raises(SystemExit, match="decode_embeddings.weight"):
        EvalRun(a, tokenizer=tok, device=cuda)


def test_eval_all_documents_mode(cuda, mini, tmp_path):
    """--eval_all_documents 1 (recall_level fine): no beam search, fine list = running top-pool over the corpus streamed
    in --encode_batch_size blocks (main_models.py:3818-3876), restated on the CPU with the oracle tower; the
    hard-negative score column is the reference's last-iteration concatenation (:3905-3908)."""
    from mevi_amd.evalrun import EvalRun, load_queries

    a = Namespace(**vars(mini["args"]))
    a.eval_all_documents, a.recall_level, a.encode_batch_size, a.recall_num, a.save_hard_neg = 1, "fine", 64, [1, 5, 10, 20, 50, 100], 150
    a.custom_save_path, a.metric_path = str(tmp_path / "all.tsv"), str(tmp_path / "m.txt")
    tok = FakeTokenizer(512)
    out = EvalRun(a, tokenizer=tok, device=cuda).run(load_queries(a.data_dir))
    prefix = a.custom_save_path[:-4]
    assert not os.path.exists(prefix + "_coarse.tsv")
    fine = [l.rstrip("\n").split("\t") for l in open(prefix + "_fine.tsv")]
    hn = [l.rstrip("\n").split("\t") for l in open(f"{prefix}_hn150.tsv")]
    enc = tok.batch_encode_plus(mini["queries"])
    qemb = ot5.tower_encode(mini["TW"], mini["tcfg"], enc["input_ids"], enc["attention_mask"])
    emb = torch.from_numpy(mini["emb"])
    N, pool = len(emb), 100
    stack = torch.empty((len(qemb), 0))
    docs = torch.empty((len(qemb), 0), dtype=torch.int64)
    for st in range(0, N, 64):                       # the reference's loop, literally
        new = qemb @ emb[st:st + 64].T
        sc = torch.cat([stack, new], -1)
        dd = torch.cat([docs, torch.arange(st, min(st + 64, N)).unsqueeze(0).expand(len(qemb), -1)], -1)
        stack, idx = torch.topk(sc, k=min(sc.shape[-1], pool), dim=-1)
        docs = dd.gather(-1, idx)
    ranks = []
    for i, q in enumerate(mini["queries"]):
        assert fine[i][0] == hn[i][0] == q and eval(fine[i][2]) == mini["gts"][i]
        got = eval(fine[i][1])
        ref_s = stack[i].numpy()
        firm = np.concatenate([[True], np.abs(np.diff(ref_s)) > 1e-4]) & np.concatenate([np.abs(np.diff(ref_s)) > 1e-4, [True]])
        assert len(got) == pool and all(got[j] == int(docs[i, j]) for j in np.nonzero(firm)[0])
        assert [int(x) for x in hn[i][2].split(",")] == got                 # pool (100) < save_hard_neg (150)
        got_sc = np.array([float(x) for x in hn[i][3].split(",")])
        want = sc[i, :150].numpy()      # 100 running-top scores + the (N % 64)-row last block, cut to 150
        assert len(got_sc) == len(want) == min(150, pool + (N - 1) % 64 + 1) and np.abs(got_sc - want).max() <= 2e-4
        ranks.append([got.index(g) if g in got else None for g in mini["gts"][i]])
    assert out["ndoc"] == N and "cluster_recall" not in out
    hit10 = np.mean([min([r for r in rk if r is not None], default=10 ** 9) < 10 for rk in ranks])
    assert abs(out["hitrate"][10] - hit10) < 1e-12 and "recallcluster10" in open(a.metric_path).read()


def _firm(ref_s, tol):
    g = np.abs(np.diff(ref_s)) > tol
    return np.concatenate([[True], g]) & np.concatenate([g, [True]])


@pytest.mark.parametrize("ratio", [0.0, 0.3])
def test_use_topic_model_ablation(cuda, mini_small, tmp_path, ratio):
    mini = mini_small
    M, K, R = 3, 4, 4
    """--use_topic_model 1 (topic_score_ratio 0; main_models.py:3539-3552): a document's score is the NCI score of its
    cluster times q.d -- on the cluster path the beam score (:3952), with --eval_all_documents the score of its code
    path among ALL K**M paths (_generate_all, generation_utils.py:1013-1136; doc2index :3311-3372), streamed through
    the running top-pool (:3818-3876).  Both against a CPU restatement from the oracle pieces, literally as the
    reference loops."""
    from mevi_amd.evalrun import EvalRun, load_queries

    tok = FakeTokenizer(512)
    enc = tok.batch_encode_plus(mini["queries"])
    ids, mask = enc["input_ids"], enc["attention_mask"]
    qemb = ot5.tower_encode(mini["TW"], mini["tcfg"], ids, mask)
    emb = torch.from_numpy(mini["emb"])
    codes_doc = orq.rq_encode(mini["emb"], mini["C"])
    cluster, _ = orq.cluster_dict(codes_doc)
    # ---- cluster path: beam score x q.d
    a = Namespace(**vars(mini["args"]))
    a.use_topic_model, a.topic_score_ratio = 1, ratio
    a.custom_save_path, a.metric_path = str(tmp_path / "t.tsv"), str(tmp_path / "m.txt")
    # all_doc_proba (gen_all_reconstruct + gen_doc2index_mapping): <reconstruct vector of the document's codes, its embedding>
    Ct = torch.from_numpy(mini["C"])
    recon = sum(Ct[j][torch.from_numpy(codes_doc[:, j].astype(np.int64))] for j in range(M))
    doc_proba = torch.sum(recon * emb, dim=-1) if ratio else 0
    EvalRun(a, tokenizer=tok, device=cuda).run(load_queries(a.data_dir))
    hn = [l.rstrip("\n").split("\t") for l in open(f"{a.custom_save_path[:-4]}_hn{a.save_hard_neg}.tsv")]
    dec, sc, _ = ot5.nci_generate(mini["W"], mini["cfg"], ids, mask, R)
    bcodes = ot5.decode_token(dec, K).view(len(ids), R, M).numpy()
    nci_scores = torch.tensor(sc.tolist(), dtype=torch.float32).reshape(len(ids), R)
    checked = 0
    for i in range(len(ids)):
        scores, docs = [], []
        for r in range(R):
            cur = cluster.get(tuple(bcodes[i, r].tolist()))
            if cur is not None:
                dp = doc_proba[cur] if ratio else 0
                scores.append(nci_scores[i][r].item() * (ratio * dp + (1 - ratio) * (qemb[i] @ emb[cur].T)))   # get_inference_scores
                docs += cur
        if not docs:
            assert hn[i][2] == ""
            continue
        ref_s, order = torch.sort(torch.cat(scores), descending=True)
        ref_d = np.array(docs)[order.numpy()]
        got_d = [int(x) for x in hn[i][2].split(",")]
        got_s = np.array([float(x) for x in hn[i][3].split(",")])
        assert sorted(got_d) == sorted(docs) and np.abs(got_s - ref_s.numpy()).max() <= 2e-4
        firm = _firm(ref_s.numpy(), 1e-3)
        assert all(got_d[j] == int(ref_d[j]) for j in np.nonzero(firm)[0])
        checked += int(firm.sum())
    assert checked > 100
    # ---- all documents: score of the document's code path among all K**M x q.d
    a = Namespace(**vars(mini["args"]))
    a.use_topic_model, a.topic_score_ratio, a.eval_all_documents, a.recall_level, a.encode_batch_size = 1, ratio, 1, "fine", 64
    a.recall_num, a.save_hard_neg = [1, 5, 10, 20, 50, 100], 150
    a.custom_save_path, a.metric_path = str(tmp_path / "ta.tsv"), str(tmp_path / "ma.txt")
    EvalRun(a, tokenizer=tok, device=cuda).run(load_queries(a.data_dir))
    fine = [l.rstrip("\n").split("\t") for l in open(a.custom_save_path[:-4] + "_fine.tsv")]
    hn = [l.rstrip("\n").split("\t") for l in open(a.custom_save_path[:-4] + "_hn150.tsv")]
    all_scores, _ = ot5.nci_generate_all(mini["W"], mini["cfg"], ids, mask)
    path = torch.from_numpy(sum(codes_doc[:, p].astype(np.int64) * K ** (M - 1 - p) for p in range(M)))
    N, pool = len(emb), 100
    stack = torch.empty((len(qemb), 0))
    docs = torch.empty((len(qemb), 0), dtype=torch.int64)
    for st in range(0, N, 64):                       # the reference's loop with use_topic_model (main_models.py:3826-3876)
        topic = all_scores[:, path[st:st + 64]]
        dp = doc_proba[st:st + 64] if ratio else 0
        new = topic * (ratio * dp + (1 - ratio) * (qemb @ emb[st:st + 64].T))
        scs = torch.cat([stack, new], -1)
        dd = torch.cat([docs, torch.arange(st, min(st + 64, N)).unsqueeze(0).expand(len(qemb), -1)], -1)
        stack, idx = torch.topk(scs, k=min(scs.shape[-1], pool), dim=-1)
        docs = dd.gather(-1, idx)
    for i in range(len(ids)):
        got = eval(fine[i][1])
        firm = _firm(stack[i].numpy(), 2e-4)
        assert len(got) == pool and all(got[j] == int(docs[i, j]) for j in np.nonzero(firm)[0]) and firm.mean() > 0.5
        got_sc = np.array([float(x) for x in hn[i][3].split(",")])
        want = scs[i, :150].numpy()
        assert len(got_sc) == len(want) and np.abs(got_sc - want).max() <= 5e-4


def test_use_topic_model_with_one_returned_sequence_weights_are_ones(cuda, mini_small, tmp_path):
    """with --num_return_sequences 1 the reference sets nci_scores = ones((B, 1)) (main_models.py:3678-3680),
    so --use_topic_model 1 ranks the single beam cluster by plain q.d -- not by (negative) hypothesis score x q.d, which
    would reverse the list."""
    from mevi_amd.evalrun import EvalRun, load_queries

    mini = mini_small
    M, K = 3, 4
    tok = FakeTokenizer(512)
    enc = tok.batch_encode_plus(mini["queries"])
    ids, mask = enc["input_ids"], enc["attention_mask"]
    qemb = ot5.tower_encode(mini["TW"], mini["tcfg"], ids, mask)
    emb = torch.from_numpy(mini["emb"])
    cluster, _ = orq.cluster_dict(orq.rq_encode(mini["emb"], mini["C"]))
    a = Namespace(**vars(mini["args"]))
    a.use_topic_model, a.topic_score_ratio, a.num_return_sequences = 1, 0.0, 1
    a.custom_save_path, a.metric_path = str(tmp_path / "t1.tsv"), str(tmp_path / "m1.txt")
    EvalRun(a, tokenizer=tok, device=cuda).run(load_queries(a.data_dir))
    hn = [l.rstrip("\n").split("\t") for l in open(f"{a.custom_save_path[:-4]}_hn{a.save_hard_neg}.tsv")]
    dec, _, _ = ot5.nci_generate(mini["W"], mini["cfg"], ids, mask, 1)
    bcodes = ot5.decode_token(dec, K).view(len(ids), 1, M).numpy()
    checked = 0
    for i in range(len(ids)):
        cur = cluster.get(tuple(bcodes[i, 0].tolist()))
        if cur is None:
            assert hn[i][2] == ""
            continue
        ref_s, order = torch.sort(1.0 * (qemb[i] @ emb[cur].T), descending=True)      # ones x q.d
        got_d = [int(x) for x in hn[i][2].split(",")]
        got_s = np.array([float(x) for x in hn[i][3].split(",")])
        assert sorted(got_d) == sorted(cur) and np.abs(got_s - ref_s.numpy()).max() <= 2e-4
        firm = _firm(ref_s.numpy(), 1e-3)
        assert all(got_d[j] == int(np.array(cur)[order.numpy()][j]) for j in np.nonzero(firm)[0])
        checked += int(firm.sum())
    assert checked > 20


def test_eval_driver_on_the_nq_dataset_path(cuda, mini, tmp_path):
    """--dataset nq_dpr: questions from nq-test.qa.csv, hits judged through test_inverse_{offsets,array}.bin; the log
    lines lose their gt columns (main_models.py:3738-3757,4060-4077).  Same model and corpus as the marco run, so
    beams and fine lists must be the ones that run logged, and the ranks must equal the reference's hit loops."""
    from mevi_amd.evalrun import EvalRun, load_nq_queries, summarize
    from mevi_amd.metrics import nq_first_hit

    a0 = mini["args"]
    prefix0 = a0.custom_save_path[:-4]
    if not os.path.exists(prefix0 + "_coarse.tsv"):
        pytest.skip("eval driver test did not run")
    rng = np.random.default_rng(9)
    nqst, N = len(mini["queries"]), mini["N"]
    fine0 = [eval(l.rstrip("\n").split("\t")[1]) for l in open(prefix0 + "_fine.tsv")]
    lists = [[] for _ in range(N)]
    for qi in range(nqst):          # answers: a doc deep in the fine list for most questions, random docs otherwise
        if fine0[qi] and qi % 4:
            lists[fine0[qi][min(len(fine0[qi]) - 1, int(rng.integers(0, 12)))]].append(qi)
        for dd in rng.choice(N, size=2, replace=False):
            lists[int(dd)].append(qi)
    data = tmp_path / "nq"
    os.makedirs(data)
    np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.int32).tofile(data / "test_inverse_offsets.bin")
    np.array([q for l in lists for q in sorted(set(l))], dtype=np.int32).tofile(data / "test_inverse_array.bin")
    lists = [sorted(set(l)) for l in lists]
    with open(data / "nq-test.qa.csv", "w") as f:
        for q in mini["queries"]:
            f.write(f"{q}\t['x']\n")
    a = Namespace(**vars(a0))
    a.dataset, a.data_dir, a.save_hard_neg = "nq_dpr", str(data), 50
    a.custom_save_path, a.metric_path = str(tmp_path / "nqres.tsv"), str(tmp_path / "nq_m.txt")
    out = EvalRun(a, tokenizer=FakeTokenizer(512), device=cuda).run(load_nq_queries(a.data_dir))
    prefix = a.custom_save_path[:-4]
    coarse = [l.rstrip("\n").split("\t") for l in open(prefix + "_coarse.tsv")]
    fine = [l.rstrip("\n").split("\t") for l in open(prefix + "_fine.tsv")]
    hn = [l.rstrip("\n").split("\t") for l in open(prefix + "_hn50.tsv")]
    coarse0 = [l.rstrip("\n").split("\t") for l in open(prefix0 + "_coarse.tsv")]
    cluster = pickle.load(open(a.pq_cluster_path, "rb"))
    offsets = np.fromfile(data / "test_inverse_offsets.bin", dtype=np.int32)
    array = np.fromfile(data / "test_inverse_array.bin", dtype=np.int32)
    res = []
    for i, q in enumerate(mini["queries"]):
        assert len(coarse[i]) == 3 and len(fine[i]) == 2 and len(hn[i]) == 4 and hn[i][1] == ""
        assert coarse[i][:2] == coarse0[i][:2] and coarse[i][2] == coarse0[i][3]       # text, beams; scores
        assert eval(fine[i][1]) == fine0[i] and [int(x) for x in hn[i][2].split(",")] == fine0[i][:50]
        cr = None
        for j, c in enumerate(eval(coarse[i][1])):           # the reference's coarse loop
            if any(i in lists[d] for d in cluster.get(tuple(c), [])):
                cr = j
                break
        res.append((q, len(fine0[i]), [cr], [nq_first_hit(i, fine0[i], offsets, array)]))
    want = summarize(res, a.recall_num, 10)
    assert out == want and want["hitrate"][100] > 0.3 and want["cluster_hitrate"][10] > 0.3


def test_offline_index_build(cuda, mini, tmp_path):
    """`main.py --mode train --only_gen_rq 1` (marco_generate_embedding_n_rq.sh): passage embeddings through per-rank part
    files, RQ codebook trained when the file is missing, cluster pickles -- then nothing is redone on a second call."""
    from mevi_amd.evalrun import load_tower_weights
    from mevi_amd.indexbuild import build_index, doc_rank_range, embed_documents
    from mevi_amd.t5 import TwinTower

    a0 = mini["args"]
    rng = np.random.default_rng(21)
    N, L = 1500, 16
    tokens = rng.integers(3, 500, size=(N, L)).astype(np.int64)
    lens = rng.integers(2, L + 1, size=N)
    masks = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    tokens[masks == 0] = 0
    tokens.tofile(tmp_path / "all_document_tokens.bin")
    masks.tofile(tmp_path / "all_document_masks.bin")
    a = Namespace(document_encoder="ance", dataset="marco", ckpt_dir=a0.ckpt_dir, document_path=str(tmp_path / "all_document"),
                  co_doc_length=L, embedding_path=str(tmp_path / "docemb.bin"), pq_path=str(tmp_path / "rqcodebook4_5.pt"),
                  pq_cluster_path=str(tmp_path / "rqclus4_5.pkl"), subvector_num=4, subvector_bits=5, encode_batch_size=64, seed=42)
    tw, dims = load_tower_weights(os.path.join(a0.ckpt_dir, "t5-ance"))
    tower = TwinTower(tw, dims=dims, device=cuda)
    want = tower.encode_passage({"input_ids": torch.from_numpy(tokens), "attention_mask": torch.from_numpy(masks)}).cpu().numpy()
    # two ranks, run one after the other: rank 1 leaves its part file, rank 0 writes its own and merges
    assert [doc_rank_range(10, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 10)]
    two = str(tmp_path / "two.bin")
    embed_documents(tower, tokens, masks, two, rank=1, nrank=2)
    embed_documents(tower, tokens, masks, two, rank=0, nrank=2)
    assert np.array_equal(np.fromfile(two, dtype=np.float32).reshape(N, -1), want)
    assert not os.path.exists(two[:-4] + "_0.bin") and not os.path.exists(two[:-4] + "_1.bin")
    n, dim, nclus = build_index(a, device=cuda)
    emb = np.fromfile(a.embedding_path, dtype=np.float32).reshape(N, dim)
    assert (n, dim) == (N, 32) and np.array_equal(emb, want)
    C = torch.load(a.pq_path, map_location="cpu").detach().numpy()
    assert C.shape == (4, 32, 32) and np.isfinite(C).all()
    cluster, mapping = orq.cluster_dict(orq.rq_encode(emb, C))           # oracle encode with the trained codebook
    got_c = pickle.load(open(a.pq_cluster_path, "rb"))
    got_m = pickle.load(open(a.pq_cluster_path.replace("clus", "mapping"), "rb"))
    assert nclus == len(got_c) and set(got_m) == set(range(N))
    differ = [d for d in range(N) if got_m[d] != mapping[d]]              # only f32-rounding near-ties may differ
    assert len(differ) <= 2 and sum(len(v) for v in got_c.values()) == N
    # the quantiser is useful: residual error well below the data's variance
    recon = sum(C[j][[got_m[d][j] for d in range(N)]] for j in range(4))
    assert ((emb - recon) ** 2).sum() < 0.6 * ((emb - emb.mean(0)) ** 2).sum()
    stamp = [os.path.getmtime(p) for p in (a.embedding_path, a.pq_path, a.pq_cluster_path)]
    assert build_index(a, device=cuda) == (n, dim, nclus)
    assert stamp == [os.path.getmtime(p) for p in (a.embedding_path, a.pq_path, a.pq_cluster_path)]


@pytest.mark.parametrize("aggr", ["add", "max"])
def test_eval_driver_with_multi_cluster_documents(cuda, mini, tmp_path, aggr):
    """--doc_multiclus 3: documents listed in the clusters of their top-3 code paths (pq.beam_search), gt codes = those
    paths, a document reached through several beams listed once with its scores summed / maxed
    (main_models.py:3222-3262,3761-3771,3997-4011), against a CPU restatement from the oracle pieces."""
    import shutil
    from collections import defaultdict

    from mevi_amd.evalrun import EvalRun, load_queries

    a = Namespace(**vars(mini["args"]))
    shutil.copy(a.pq_path, tmp_path / "rqcodebook4_5.pt")
    a.pq_path, a.pq_cluster_path = str(tmp_path / "rqcodebook4_5.pt"), str(tmp_path / "rqclus4_5.pkl")
    a.custom_save_path, a.metric_path = str(tmp_path / "mc.tsv"), str(tmp_path / "mc_m.txt")
    a.doc_multiclus, a.multiclus_score_aggr = 3, aggr
    tok = FakeTokenizer(512)
    out = EvalRun(a, tokenizer=tok, device=cuda).run(load_queries(a.data_dir))
    labels, _ = orq.rq_beam_search(mini["emb"], mini["C"], 3)
    got_labels = torch.load(str(tmp_path / "rqtopk34_5.pt")).numpy()
    assert got_labels.shape == labels.shape and (got_labels != labels).any(axis=(1, 2)).mean() < 0.01   # near-tie paths only
    multi = defaultdict(list)
    for i, paths in enumerate(got_labels.tolist()):
        for p_ in paths:
            multi[tuple(p_)].append(i)
    assert pickle.load(open(tmp_path / "rqmulticlus34_5.pkl", "rb")) == dict(multi)
    enc = tok.batch_encode_plus(mini["queries"])
    dec, _, _ = ot5.nci_generate(mini["W"], mini["cfg"], enc["input_ids"], enc["attention_mask"], 10)
    codes = ot5.decode_token(dec, 32).view(len(mini["queries"]), 10, 4).numpy()
    qemb = ot5.tower_encode(mini["TW"], mini["tcfg"], enc["input_ids"], enc["attention_mask"]).numpy()
    prefix = a.custom_save_path[:-4]
    coarse = [l.rstrip("\n").split("\t") for l in open(prefix + "_coarse.tsv")]
    hn = [l.rstrip("\n").split("\t") for l in open(f"{prefix}_hn{a.save_hard_neg}.tsv")]
    nd, repeats = 0, 0
    for i, q in enumerate(mini["queries"]):
        d = codes[i].tolist()
        assert eval(coarse[i][1]) == d
        gt_paths = [got_labels[g].tolist() for g in mini["gts"][i]]
        assert eval(coarse[i][2]) == gt_paths
        docs = [x for c in d for x in multi.get(tuple(c), [])]
        nd += len(docs)
        if not docs:
            continue
        u, cnt = np.unique(docs, return_counts=True)
        repeats += int((cnt > 1).sum())
        base = mini["emb"][u] @ qemb[i]
        want = base * cnt if aggr == "add" else base
        got_docs = [int(x) for x in hn[i][2].split(",")]
        got_s = np.array([float(x) for x in hn[i][3].split(",")])
        assert sorted(got_docs) == u.tolist()
        order = np.argsort(-want, kind="stable")
        assert np.abs(got_s - want[order]).max() <= 5e-4
        gaps = np.abs(np.diff(want[order]))
        firm = np.concatenate([[True], gaps > 2e-3]) & np.concatenate([gaps > 2e-3, [True]])
        assert all(got_docs[j] == int(u[order[j]]) for j in np.nonzero(firm)[0])
    assert repeats > 5, "fixture should reach some documents through several beams"
    assert abs(out["ndoc"] - nd / len(mini["queries"])) < 1e-9


@pytest.mark.parametrize("aggr,ratio", [("add", 0.0), ("max", 0.3)])
def test_topic_model_over_multi_cluster_documents(cuda, mini, tmp_path, aggr, ratio):
    """--use_topic_model 1 --doc_multiclus 3 on the cluster path (main_models.py:3944-4011 with :3539-3552 and the per-(document,
    cluster) probability of gen_doc2index_mapping :3311-3372): a candidate reached through beam cluster i scores
    nci_score_i * (ratio * <reconstruct(cluster i), emb[d]> + (1 - ratio) * q.d); a document reached through several beams is
    listed once with those scores summed in list order / maxed.  Against the reference's loop restated from oracle pieces."""
    import shutil
    from collections import defaultdict

    from mevi_amd.evalrun import EvalRun, load_queries

    a = Namespace(**vars(mini["args"]))
    shutil.copy(a.pq_path, tmp_path / "rqcodebook4_5.pt")
    a.pq_path, a.pq_cluster_path = str(tmp_path / "rqcodebook4_5.pt"), str(tmp_path / "rqclus4_5.pkl")
    a.custom_save_path, a.metric_path = str(tmp_path / "tm.tsv"), str(tmp_path / "tm_m.txt")
    a.doc_multiclus, a.multiclus_score_aggr, a.use_topic_model, a.topic_score_ratio = 3, aggr, 1, ratio
    tok = FakeTokenizer(512)
    EvalRun(a, tokenizer=tok, device=cuda).run(load_queries(a.data_dir))
    got_labels = torch.load(str(tmp_path / "rqtopk34_5.pt")).numpy()
    multi = defaultdict(list)
    for i, paths in enumerate(got_labels.tolist()):
        for p_ in paths:
            multi[tuple(p_)].append(i)
    enc = tok.batch_encode_plus(mini["queries"])
    dec, sc, _ = ot5.nci_generate(mini["W"], mini["cfg"], enc["input_ids"], enc["attention_mask"], 10)
    codes = ot5.decode_token(dec, 32).view(len(mini["queries"]), 10, 4).numpy()
    nci_scores = torch.tensor(sc.tolist(), dtype=torch.float32).reshape(len(mini["queries"]), 10)
    qemb = ot5.tower_encode(mini["TW"], mini["tcfg"], enc["input_ids"], enc["attention_mask"])
    emb, Ct = torch.from_numpy(mini["emb"]), torch.from_numpy(mini["C"])
    hn = [l.rstrip("\n").split("\t") for l in open(f"{a.custom_save_path[:-4]}_hn{a.save_hard_neg}.tsv")]
    checked = merged = 0
    for i in range(len(mini["queries"])):
        scores, docs = [], []
        for r in range(10):
            cur = multi.get(tuple(codes[i, r].tolist()))
            if cur is None:
                continue
            recon = sum(Ct[j][int(codes[i, r, j])] for j in range(4))
            dp = (emb[cur] @ recon) if ratio else 0
            scores.append(nci_scores[i][r].item() * (ratio * dp + (1 - ratio) * (qemb[i] @ emb[cur].T)))
            docs += cur
        if not docs:
            assert hn[i][2] == ""
            continue
        scores, docs = torch.cat(scores), np.array(docs)
        udocs, uidx = np.unique(docs, return_inverse=True)
        us = torch.zeros(len(udocs)) if aggr == "add" else torch.full((len(udocs),), -float("inf"))
        for ui, s_ in zip(uidx, scores):                            # main_models.py:4003-4008
            us[ui] = us[ui] + s_ if aggr == "add" else torch.max(us[ui], s_)
        merged += len(docs) - len(udocs)
        ref_s, order = torch.sort(us, descending=True)
        got_d = [int(x) for x in hn[i][2].split(",")]
        got_s = np.array([float(x) for x in hn[i][3].split(",")])
        assert sorted(got_d) == udocs.tolist() and np.abs(got_s - ref_s.numpy()).max() <= 5e-4 * max(1.0, float(ref_s.abs().max()))
        firm = _firm(ref_s.numpy(), 2e-3)
        assert all(got_d[j] == int(udocs[order.numpy()][j]) for j in np.nonzero(firm)[0])
        checked += int(firm.sum())
    assert checked > 50 and merged > 5


def test_dense_cli_and_ensemble_chain(cuda, mini, tmp_path):
    """faiss_search.py (C1-style plumbing on the GPU) -> evaluate.py -> ensemble_marco.py on the files above."""
    d, a = mini["dir"], mini["args"]
    enc = FakeTokenizer(512).batch_encode_plus(mini["queries"])
    from mevi_amd.t5 import TwinTower
    from mevi_amd.evalrun import load_tower_weights

    tw, dims = load_tower_weights(os.path.join(a.ckpt_dir, "t5-ance"))
    q = TwinTower(tw, dims=dims, device=cuda).encode_query(enc).cpu().numpy()
    q.tofile(d / "ance" / "query_emb.bin")
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = str(d / "ance" / "dense.txt")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "faiss_search.py"), "--query_path", str(d / "ance" / "query_emb.bin"),
                        "--doc_path", a.embedding_path, "--output_path", out, "--raw_query_path",
                        str(d / "origin" / "dev_mevi_dedup.tsv"), "--dim", "32", "--topk", "100", "--param", "HNSW256"],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "Param HNSW256 trained: True." in r.stdout and "int64 (23, 100) float32 (23, 100)" in r.stdout
    es, ei = odense.ip_topk_exact(q, mini["emb"], 100)
    lines = [l.rstrip("\n").split("\t") for l in open(out)]
    for i, l in enumerate(lines):
        assert l[0] == mini["queries"][i] and l[1] == ""
        assert [int(x) for x in l[2].split(",")] == ei[i].tolist()
        assert [np.float32(x) for x in l[3].split(",")] == es[i].tolist()       # bit-exact scores through the TSV
    prefix = a.custom_save_path[:-4]
    if not os.path.exists(prefix + "_coarse.tsv"):
        pytest.skip("eval driver test did not run")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "ensemble_marco.py"), "--mapping_file",
                        a.pq_cluster_path.replace("clus", "mapping"), "--gt_file", str(d / "origin" / "dev_mevi_dedup.tsv"),
                        "--ance_file", out, "--coarse_file", prefix + "_coarse.tsv", "--fine_file",
                        f"{prefix}_hn{a.save_hard_neg}.tsv", "--ofile", str(tmp_path / "ens.txt")],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "ANCE Pred" in r.stdout and "Fine Pred" in r.stdout and "score + 0.6 / (0.03 * crank + 1)" in r.stdout


def test_command_lines_with_a_real_sentencepiece_tokenizer(cuda, mini, tmp_path):
    """generate.py --gen_query and main.py --mode eval through their REAL command lines with no injected tokenizer: both load
    `AutoTokenizer.from_pretrained(<ckpt>/t5-ance)` (a SentencePiece model trained for the test, tests/spm_fixture.py) from
    the installed `transformers` -- whose tokenizers no longer have the reference's `batch_encode_plus`.  Embeddings equal
    the oracle tower on the tokenizer's own ids; the eval run's coarse log carries the oracle's beams for those ids."""
    pytest.importorskip("sentencepiece")
    from transformers import AutoTokenizer

    from spm_fixture import build_t5_tokenizer_dir

    a0 = mini["args"]
    ck = tmp_path / "ckpts"
    shutil.copytree(a0.ckpt_dir, ck)
    build_t5_tokenizer_dir(str(ck / "t5-ance"))
    tok = AutoTokenizer.from_pretrained(str(ck / "t5-ance"))
    enc = tok(mini["queries"], max_length=32, padding="max_length", truncation=True, return_tensors="pt")
    assert int(enc["input_ids"].max()) < 512                              # inside the miniature models' vocabulary
    env = dict(os.environ, PYTHONPATH=ROOT)
    qfile = os.path.join(a0.data_dir, "dev_mevi_dedup.tsv")
    out = str(tmp_path / "query_emb.bin")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "generate.py"), "--query_file", qfile, "--model_path", str(ck / "t5-ance"),
                        "--tokenizer_path", str(ck / "t5-ance"), "--query_embedding_path", out, "--dim", "32", "--gpus", "0",
                        "--gen_query"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.fromfile(out, dtype=np.float32).reshape(-1, 32)
    want = ot5.tower_encode(mini["TW"], mini["tcfg"], enc["input_ids"], enc["attention_mask"]).numpy()
    assert got.shape == want.shape and np.abs(got - want).max() <= 5e-5
    save = str(tmp_path / "res" / "nci_result.tsv")
    os.makedirs(tmp_path / "res")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "--mode", "eval", "--n_gpu", "1", "--codebook", "1", "--pq_type", "rq",
                        "--subvector_num", "4", "--subvector_bits", "5", "--query_encoder", "twin", "--document_encoder", "ance",
                        "--recall_level", "both", "--num_return_sequences", "10", "--adaptor_layer_num", "2", "--eval_batch_size", "2",
                        "--nci_ckpt", a0.nci_ckpt, "--ckpt_dir", str(ck), "--data_dir", a0.data_dir, "--embedding_path", a0.embedding_path,
                        "--pq_path", a0.pq_path, "--pq_cluster_path", a0.pq_cluster_path, "--custom_save_path", save,
                        "--save_hard_neg", str(a0.save_hard_neg), "--logs_dir", str(tmp_path / "logs"), "--learning_rate", "2e-4",
                        "--fixnci", "--fixpq"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    coarse = [l.rstrip("\n").split("\t") for l in open(save[:-4] + "_coarse.tsv")]
    dec, sc, _ = ot5.nci_generate(mini["W"], mini["cfg"], enc["input_ids"], enc["attention_mask"], 10)
    codes = ot5.decode_token(dec, 32).view(len(mini["queries"]), 10, 4).numpy()
    assert len(coarse) == len(mini["queries"])
    for i, q in enumerate(mini["queries"]):
        assert coarse[i][0] == q and eval(coarse[i][1]) == codes[i].tolist()
        assert np.abs(np.array(eval(coarse[i][3])) - sc.numpy().reshape(-1, 10)[i]).max() <= 1e-5
    # the latency hooks of the two scripts (generate.py:245-281 `--timing_infer_step N` -> timer.pkl, N entries at batch 1;
    # main_models.py:3558,3729-3732,4057-4059,4091-4096 -> times<R>.pkl with N + 1 steps of eval_batch_size queries, then exit)
    import pickle

    wd = tmp_path / "timing"
    os.makedirs(wd)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "generate.py"), "--query_file", qfile, "--model_path", str(ck / "t5-ance"),
                        "--tokenizer_path", str(ck / "t5-ance"), "--dim", "32", "--gpus", "0", "--timing_infer_step", "5"],
                       capture_output=True, text=True, env=env, cwd=wd)
    assert r.returncode == 0, r.stderr[-2000:]
    timer = pickle.load(open(wd / "timer.pkl", "rb"))
    assert len(timer) == 5 and all(0 < t < 5 for t in timer)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "--mode", "eval", "--n_gpu", "1", "--codebook", "1", "--pq_type", "rq",
                        "--subvector_num", "4", "--subvector_bits", "5", "--query_encoder", "twin", "--document_encoder", "ance",
                        "--recall_level", "both", "--num_return_sequences", "10", "--adaptor_layer_num", "2", "--eval_batch_size", "2",
                        "--nci_ckpt", a0.nci_ckpt, "--ckpt_dir", str(ck), "--data_dir", a0.data_dir, "--embedding_path", a0.embedding_path,
                        "--pq_path", a0.pq_path, "--pq_cluster_path", a0.pq_cluster_path, "--custom_save_path", str(wd / "t.tsv"),
                        "--logs_dir", str(tmp_path / "logs"), "--fixnci", "--fixpq", "--timing_infer_step", "3"],
                       capture_output=True, text=True, env=env, cwd=wd)
    assert r.returncode == 0, r.stderr[-2000:]
    times = pickle.load(open(wd / "times10.pkl", "rb"))
    assert sorted(times) == ["knn", "nci"] and len(times["nci"]) == len(times["knn"]) == 4
    assert all(0 < t < 5 for t in times["nci"] + times["knn"])


def test_eval_driver_with_a_bert_tower(cuda, mini, tmp_path):
    """--document_encoder cocondenser: the fine stage scores with a BERT-family tower that reads the query through its
    own tokenizer (no special tokens for 'cocondenser'), in the tower's embedding space (48-d here, NCI is 32-d)."""
    from mevi_amd.evalrun import EvalRun, load_queries
    from oracle import bert as obert

    g = np.load(os.path.join(GOLD, "g8_bert_tower.npz"))
    bcfg = json.loads(str(g["cfg"]))
    BW = obert.load_weights(g)
    d = tmp_path
    a0 = mini["args"]
    ck = d / "ckpts"
    os.makedirs(ck / "co-condenser-marco-retriever")
    os.symlink(os.path.join(a0.ckpt_dir, "t5-ance"), ck / "t5-ance")
    torch.save(BW, ck / "co-condenser-marco-retriever" / "pytorch_model.bin")
    json.dump(dict(model_type="bert", **bcfg), open(ck / "co-condenser-marco-retriever" / "config.json", "w"))
    rng = np.random.default_rng(5)
    dim, M, K, N = 48, 4, 32, mini["N"]          # at least as many docs as the gt ids of the query file refer to
    C = (rng.standard_normal((M, K, dim)) * (1.0 / np.arange(1, M + 1))[:, None, None]).astype(np.float32)
    tok = FakeTokenizer(512)
    enc = tok.batch_encode_plus(mini["queries"])
    dec, _, _ = ot5.nci_generate(mini["W"], mini["cfg"], enc["input_ids"], enc["attention_mask"], 10)
    beam_codes = ot5.decode_token(dec, K).numpy()
    paths = np.concatenate([np.repeat(beam_codes[::3], 4, axis=0), rng.integers(0, K, size=(N, M))])
    emb = sum(C[j][paths[:, j]] for j in range(M)).astype(np.float32) + 0.01 * rng.standard_normal((len(paths), dim)).astype(np.float32)
    os.makedirs(d / "ance")
    emb.tofile(d / "ance" / "docemb.bin")
    torch.save(torch.nn.Parameter(torch.from_numpy(C)), d / "ance" / "rqcodebook4_5.pt")
    args = Namespace(**{**vars(a0), **dict(ckpt_dir=str(ck), embedding_path=str(d / "ance" / "docemb.bin"),
                                           pq_path=str(d / "ance" / "rqcodebook4_5.pt"),
                                           pq_cluster_path=str(d / "ance" / "rqclus4_5.pkl"),
                                           custom_save_path=str(d / "ance" / "res.tsv"), save_hard_neg=len(paths),
                                           metric_path=str(d / "logs" / "m.txt"), document_encoder="cocondenser",
                                           dataset="marco")})

    class BertFake(FakeTokenizer):
        def batch_encode_plus(self, texts, max_length=32, padding="max_length", truncation=True, return_tensors="pt",
                              add_special_tokens=True):
            assert add_special_tokens is False            # 'cocondenser' (main_models.py:359-360)
            return super().batch_encode_plus(texts, max_length=max_length)

    btok = BertFake(400)
    run = EvalRun(args, tokenizer=tok, device=cuda, tower_tokenizer=btok)
    run.run(load_queries(args.data_dir))
    prefix = args.custom_save_path[:-4]
    hn = [l.rstrip("\n").split("\t") for l in open(f"{prefix}_hn{args.save_hard_neg}.tsv")]
    fine = [l.rstrip("\n").split("\t") for l in open(prefix + "_fine.tsv")]
    benc = FakeTokenizer(400).batch_encode_plus(mini["queries"])
    qemb = obert.tower_encode(BW, bcfg, benc["input_ids"], benc["attention_mask"]).numpy()
    cluster, _ = orq.cluster_dict(orq.rq_encode(emb, C))
    codes = beam_codes.reshape(len(mini["queries"]), 10, 4)
    checked = 0
    for i in range(len(mini["queries"])):
        docs = [x for c in codes[i].tolist() for x in cluster.get(tuple(c), [])]
        got_docs = eval(fine[i][1])
        assert sorted(got_docs) == sorted(docs)
        if docs:
            ref = np.sort(emb[docs] @ qemb[i])[::-1]
            got_s = np.array([float(x) for x in hn[i][3].split(",")])
            assert np.abs(got_s - ref).max() <= 3e-4
            checked += 1
    assert checked >= 5


def test_cli_level_rehearsal_at_reduced_size(cuda, tmp_path):
    """tools/e2e_fullsize.py with 300 k documents: t5-base-shaped checkpoints on disk -> EvalRun start-up (pinned corpus
    upload, RQ encode, pickles), eval run with its logs, tower + dense search + TSV, ensemble consumer -- the flow the
    MS MARCO-sized profile (profiles/r01_e2e_fullsize.txt) times, kept runnable."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_fullsize.py"), str(tmp_path / "scratch"), "300000"],
                       capture_output=True, text=True, env=dict(os.environ, PYTHONPATH=ROOT), timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "TOTAL" in r.stdout and "ensemble_marco.py equivalent" in r.stdout and "dense.txt" in r.stdout
    assert not os.path.exists(tmp_path / "scratch")          # the scratch directory is removed at the end


def test_pinned_upload_equals_plain_copy(cuda):
    from mevi_amd.io import upload_rows

    rng = np.random.default_rng(4)
    for rows, chunk in ((0, 8), (5, 8), (1000, 64), (1000, 1000), (1001, 250)):
        a = rng.standard_normal((rows, 24)).astype(np.float32)
        assert torch.equal(upload_rows(a, cuda, chunk_rows=chunk).cpu(), torch.from_numpy(a))


def test_pinned_upload_of_a_mapped_file_reads_straight_into_the_staging_buffers(cuda, tmp_path):
    """faiss_search.read hands upload_rows a file-backed memmap: the rows are pread into the pinned buffers (no user-space
    copy of the mapping) -- same tensor as the plain copy, for a mapping that starts inside the file and ragged chunks."""
    from mevi_amd.io import map_rows, upload_rows

    rng = np.random.default_rng(5)
    a = rng.standard_normal((1003, 24)).astype(np.float32)
    a.tofile(tmp_path / "rows.bin")
    for first, rows, chunk in ((0, None, 250), (17, 500, 64), (1000, 3, 8)):
        m = map_rows(str(tmp_path / "rows.bin"), 24, rows=rows, first_row=first)
        want = a[first:first + (rows if rows is not None else len(a))]
        assert isinstance(m, np.memmap) and torch.equal(upload_rows(m, cuda, chunk_rows=chunk).cpu(), torch.from_numpy(want))
    # SLICES of a mapping (numpy keeps the root's .offset on them: the file position has to come from the addresses), more chunks
    # than staging buffers, fewer, and the default chunk size
    whole = map_rows(str(tmp_path / "rows.bin"), 24)
    inner = map_rows(str(tmp_path / "rows.bin"), 24, first_row=3)
    for view, want, chunk in ((whole[17:517], a[17:517], 64), (inner[5:9], a[8:12], 8), (whole[1:], a[1:], 100), (whole[900:], a[900:], None),
                              (np.memmap(str(tmp_path / "rows.bin"), dtype=np.float32, mode="r").reshape(-1, 24)[10:20], a[10:20], 3)):
        assert torch.equal(upload_rows(view, cuda, chunk_rows=chunk).cpu(), torch.from_numpy(np.ascontiguousarray(want)))
        assert torch.equal(upload_rows(view, cuda, chunk_rows=chunk, buffers=2, threads=3).cpu(), torch.from_numpy(np.ascontiguousarray(want)))


def _two_rank_eval_worker(rank, world, port, args_dict, ret):
    import torch.distributed as dist

    from mevi_amd.evalrun import EvalRun, load_queries

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a = Namespace(**args_dict)
    if rank == 1 and getattr(a, "late_rank_s", 0):
        import time

        time.sleep(a.late_rank_s)      # a rank that reaches the pickle check seconds after rank 0 (27 GB uploads skew ranks)
    out = EvalRun(a, tokenizer=FakeTokenizer(512), rank=rank, nrank=world, barrier=dist.barrier,
                  device=torch.device("cuda:0")).run(load_queries(a.data_dir))
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("first_run", [False, True])
def test_eval_driver_with_two_ranks(cuda, mini, tmp_path, first_run):
    """main.py --n_gpu 2 in miniature (two processes sharing this GPU, gloo barrier): every rank takes its
    DistributedSampler slice, the per-rank logs are merged in rank order (the sampler's padding repeats a head sample),
    rank 0 aggregates the metrics from all ranks' results -- same lines and same metrics as the single-rank run.
    first_run: the cluster pickles do not exist yet and rank 1 arrives seconds late -- every rank must take the same
    branch around rank 0's write (they decide before anyone writes), or the barriers pair off by one."""
    import socket

    import torch.multiprocessing as mp

    a0 = mini["args"]
    prefix0 = a0.custom_save_path[:-4]
    if not os.path.exists(prefix0 + "_coarse.tsv"):
        pytest.skip("eval driver test did not run")
    a = dict(vars(a0))
    a["custom_save_path"], a["metric_path"] = str(tmp_path / "two.tsv"), str(tmp_path / "two_m.txt")
    if first_run:
        a["pq_cluster_path"], a["late_rank_s"] = str(tmp_path / "rqclus4_5.pkl"), 4
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_two_rank_eval_worker, nprocs=2, args=(2, port, a, ret))
    assert ret[1] is None and ret[0] is not None
    prefix = a["custom_save_path"][:-4]
    n = len(mini["queries"])                      # 23 queries -> 12 per rank, the last one of rank 1 repeats query 0
    for suffix in ("_coarse.tsv", "_fine.tsv", f"_hn{a0.save_hard_neg}.tsv"):
        one = open(prefix0 + suffix).read().splitlines()
        two = open(prefix + suffix).read().splitlines()
        assert len(two) == n + 1 and set(two) == set(one)
        assert two[:12] == one[0::2] and two[12:23] == one[1::2] and two[23] == one[0]
    assert open(a["metric_path"]).read() == open(a0.metric_path).read()
    assert ret[0]["nqueries"] == n
    if first_run:
        assert pickle.load(open(a["pq_cluster_path"], "rb")) == pickle.load(open(a0.pq_cluster_path, "rb"))


def test_generate_py_gen_doc_cli_with_two_ranks(cuda, mini, tmp_path):
    """`generate.py --gen_doc --gpus 0,0` as a real command line (mp.spawn, one process per entry of --gpus, part files
    merged by rank 0; MEVI_DIST_BACKEND=gloo because both ranks sit on this one GPU) against `--gpus 0`."""
    rng = np.random.default_rng(8)
    N, L = 301, 16
    tokens = rng.integers(3, 500, size=(N, L)).astype(np.int64)
    lens = rng.integers(2, L + 1, size=N)
    masks = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    tokens[masks == 0] = 0
    tokens.tofile(tmp_path / "all_document_tokens.bin")
    masks.tofile(tmp_path / "all_document_masks.bin")
    model_dir = os.path.join(mini["args"].ckpt_dir, "t5-ance")
    outs = []
    for gpus in ("0", "0,0"):
        out = str(tmp_path / f"emb_{len(gpus)}.bin")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "generate.py"), "--model_path", model_dir, "--tokenizer_path", model_dir,
                            "--gen_doc", "--document_dir", str(tmp_path), "--doc_embedding_path", out, "--gpus", gpus, "--dim", "32",
                            "--doc_length", str(L), "--batch_size", "64"],
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, PYTHONPATH=ROOT, MEVI_DIST_BACKEND="gloo"))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.fromfile(out, dtype=np.float32).reshape(N, 32))
    assert np.array_equal(outs[0], outs[1]) and np.isfinite(outs[0]).all() and np.abs(outs[0]).max() > 0
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("emb_")) == ["emb_1.bin", "emb_3.bin"]


def test_faiss_search_py_under_torchrun_with_two_ranks(cuda, mini, tmp_path):
    """`torchrun --nproc-per-node 2 faiss_search.py ...`: the corpus file is row-sharded over the ranks, every rank
    searches its rows with global ids, rank 0 writes the merged TSV -- byte-identical to the single-process run."""
    import socket

    d, a = mini["dir"], mini["args"]
    q = np.random.default_rng(2).standard_normal((len(mini["queries"]), 32)).astype(np.float32)
    q.tofile(tmp_path / "q.bin")
    base = [os.path.join(ROOT, "faiss_search.py"), "--query_path", str(tmp_path / "q.bin"), "--doc_path", a.embedding_path,
            "--raw_query_path", str(d / "origin" / "dev_mevi_dedup.tsv"), "--dim", "32", "--topk", "300", "--param", "Flat"]
    env = dict(os.environ, PYTHONPATH=ROOT, MEVI_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable] + base + ["--output_path", str(tmp_path / "one.txt")], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-1500:]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port)] + base + ["--output_path", str(tmp_path / "two.txt")],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    assert open(tmp_path / "one.txt", "rb").read() == open(tmp_path / "two.txt", "rb").read()


def test_corpus_upload_overlapped_with_the_model_loads(cuda, mini, monkeypatch):
    """Round 6: EvalRun starts the corpus upload on a background thread (its own stream) before it loads the checkpoints and
    joins it where the embeddings are first needed (files >= 256 MB; MEVI_OVERLAP_UPLOAD=always forces it here).  The resident
    matrix must be the file, and a run must log what the serial start-up logs."""
    from mevi_amd.evalrun import EvalRun, _corpus_width_hint, load_queries

    a = mini["args"]
    monkeypatch.setenv("MEVI_OVERLAP_UPLOAD", "0")
    assert _corpus_width_hint(a) is None
    serial = EvalRun(a, tokenizer=FakeTokenizer(512), device=cuda)
    want = serial.run(load_queries(a.data_dir))
    monkeypatch.setenv("MEVI_OVERLAP_UPLOAD", "always")
    assert _corpus_width_hint(a) == serial.tower.dim
    over = EvalRun(a, tokenizer=FakeTokenizer(512), device=cuda)
    assert torch.equal(over.emb, serial.emb)
    emb_file = np.fromfile(a.embedding_path, dtype=np.float32).reshape(-1, serial.tower.dim)
    assert np.array_equal(over.emb.cpu().numpy(), emb_file)
    got = over.run(load_queries(a.data_dir))
    assert got["recall"] == want["recall"] and got["ndoc"] == want["ndoc"]
