#!/usr/bin/env python3
"""Capture golden vectors by importing the read-only reference (/root/reference).

Runs ONLY in the build container.  Everything written to tests/golden/ is data
(seeded inputs + the reference's outputs); no reference source travels.

  python tools/capture_goldens.py [g4 g5 g6 g7 g1 g2 g3 ...]   (default: all)
"""
import io
import json
import os
import pickle
import shutil
import subprocess
import sys
import tempfile
from argparse import Namespace
from contextlib import redirect_stdout

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)

import ref_import  # noqa: E402

REF = ref_import.REF


# --------------------------------------------------------------------------- G4
def g4_rq():
    """RQ encode / forward / beam search of MEVI/pq.py (ProductQuantization('rq', M, bits, 'l2'))."""
    ref_import.setup()
    import torch
    from pq import ProductQuantization

    for (M, bits, dim, n) in [(4, 5, 768, 192), (3, 8, 64, 300), (4, 5, 32, 257)]:
        K = 2 ** bits
        rng = np.random.default_rng(100 + M * 10 + bits)
        X = rng.standard_normal((n, dim)).astype(np.float32)
        C = (rng.standard_normal((M, K, dim)) * (0.8 / np.arange(1, M + 1))[:, None, None]).astype(np.float32)
        pq = ProductQuantization("rq", M, bits, "l2", dim, pq_init_method="none", pq_update_method="none")
        with torch.no_grad():
            pq.codebook.copy_(torch.from_numpy(C))
        pq.eval()
        with io.StringIO() as buf, redirect_stdout(buf):
            cluster, mapping = pq.get_document_cluster(X, 0, 1, batch_size=128, return_mapping=True)
        codes = np.array([mapping[i] for i in range(n)], dtype=np.int32)
        proba, index, _ = pq.forward(torch.from_numpy(X.copy()), return_loss=False)
        assert np.array_equal(index.numpy(), codes)
        out = dict(X=X, C=C, codes=codes, forward_proba=proba.numpy().astype(np.float32))
        for R in (5, 10):
            if R > K:
                continue
            lab, sc = pq.beam_search(torch.from_numpy(X[:64].copy()), R, return_proba=True)
            out[f"beam{R}_labels"] = lab.numpy().astype(np.int32)
            out[f"beam{R}_scores"] = sc.numpy().astype(np.float32)
        rec = torch.stack([pq.get_reconstruct_vector(torch.from_numpy(codes[r].astype(np.int64)))
                           for r in range(32)])  # 1-D index per row (the only form infer() uses, main_models.py:3938)
        out["reconstruct32"] = rec.detach().numpy().astype(np.float32)
        # cluster dict in a flat form: sorted (code tuple -> doc ids)
        keys = sorted(cluster)
        out["cluster_keys"] = np.array(keys, dtype=np.int32)
        out["cluster_sizes"] = np.array([len(cluster[k]) for k in keys], dtype=np.int32)
        out["cluster_docs"] = np.array([d for k in keys for d in cluster[k]], dtype=np.int64)
        np.savez_compressed(os.path.join(GOLD, f"g4_rq_{M}_{bits}_{dim}.npz"), **out)
        print("g4", M, bits, dim, "clusters", len(keys))


# --------------------------------------------------------------------------- G5
def g5_tree_codec():
    """Token codec + shared-layer prefix tree (main_models.TreeBuilder / encode_single_newid / decode_token)."""
    ref_import.setup()
    import torch
    from main_models import TreeBuilder, decode_token, encode_single_newid
    from main_utils import dec_2d

    cases = []
    for (M, K) in [(4, 32), (3, 256), (3, 16), (2, 4)]:
        args = Namespace(kary=K, position=1, label_length_cutoff=M, max_output_length=M + 2,
                         codebook=1, subvector_num=M, output_vocab_size=K)
        rng = np.random.default_rng(M * 1000 + K)
        codes = rng.integers(0, K, size=(12, M))
        enc = [encode_single_newid(args, list(map(int, c))) for c in codes]
        enc_str = encode_single_newid(args, "-".join(str(int(x)) for x in codes[0]))
        # tree exactly as T5FineTuner.build_tree (main_models.py:1698-1706)
        builder = TreeBuilder(share_sons=True)
        newids = [encode_single_newid(args, [i for _ in range(M)]) for i in range(K)]
        for i in range(M):
            builder.add_layer([ids[i] for ids in newids])
        builder.add_layer([1])
        root = builder.build()
        levels = []
        node = root
        while node.children:
            levels.append(sorted(node.children.keys()))
            node = next(iter(node.children.values()))
        seqs = torch.tensor([[0] + e for e in enc], dtype=torch.long)
        dec, eos = decode_token(args, seqs.clone())
        d2 = dec_2d(dec, 3)
        cases.append(dict(M=M, K=K, codes=codes.tolist(), encoded=enc, encoded_from_str=enc_str,
                          tree_levels=levels, decoded=dec.tolist(), eos_is_none=eos is None,
                          dec_2d_shape=list(d2.shape), dec_2d=d2.tolist()))
    json.dump(cases, open(os.path.join(GOLD, "g5_tree_codec.json"), "w"))
    print("g5", len(cases), "cases")


# --------------------------------------------------------------------------- G6
def synth_consumer_files(d, nq=60, ndoc=3000, kd=40, R=10, M=4, K=8, seed=0):
    """Synthetic gt / dense / coarse / hn TSVs + mapping pkl in the reference's formats
    (formats: SURVEY 8(b); writers: faiss_search.py:71-77, main_models.py:3760,4046-4053)."""
    rng = np.random.default_rng(seed)
    mapping = {i: tuple(int(x) for x in rng.integers(0, K, size=M)) for i in range(ndoc)}
    with open(os.path.join(d, "rqmapping.pkl"), "wb") as f:
        pickle.dump(mapping, f)
    queries = [f"what is query number {i} about" for i in range(nq)]
    gt, dense, coarse, hn = [], [], [], []
    for i, q in enumerate(queries):
        ngt = 1 + (i % 3 == 0)
        g = [int(x) for x in rng.choice(ndoc, size=ngt, replace=False)]
        gt.append(f"{q}\t{','.join(map(str, g))}")
        # dense list: kd ids with descending scores; plant gt at a pseudo-random rank for 2/3 of queries
        ids = [int(x) for x in rng.choice(ndoc, size=kd, replace=False) if x not in g][:kd - 2]
        if i % 3 != 1:
            ids.insert(int(rng.integers(0, min(len(ids), 25))), g[0])
        while len(ids) < kd:
            ids.append(-1 if i % 7 == 0 else int(rng.integers(0, ndoc)))  # faiss pads with -1 when k > N
        sc = np.sort(rng.normal(80, 3, size=kd).astype(np.float32))[::-1]
        if i % 7 == 0:
            sc[np.array(ids) == -1] = -3.4028234663852886e+38
        dense.append(f"{q}\t\t{','.join(str(x) for x in ids)}\t{','.join(str(float(s)) for s in sc)}")
        # coarse: R distinct clusters; first ones drawn from the dense/gt docs so ranks are non-trivial
        clus = []
        cand = [mapping[g[0]]] if i % 4 != 3 else []
        cand += [mapping[x] for x in ids[:15] if x >= 0]
        for c in cand:
            if c not in clus:
                clus.append(c)
        while len(clus) < R:
            c = tuple(int(x) for x in rng.integers(0, K, size=M))
            if c not in clus:
                clus.append(c)
        order = rng.permutation(len(clus))[:R]
        clus = [list(clus[j]) for j in order]
        cs = [float(x) for x in np.sort(rng.normal(-2, 0.5, size=R))[::-1]]
        coarse.append(f"{q}\t{clus}\t{[list(mapping[x]) for x in g]}\t{cs}")
        # hn (fine) file: docs that live in the beam clusters, sorted by score; some overlap with dense
        members = [x for x in range(ndoc) if list(mapping[x]) in clus]
        rng.shuffle(members)
        fdocs = members[: int(rng.integers(5, 60))]
        fs = np.sort(rng.normal(79, 3, size=len(fdocs)).astype(np.float32))[::-1]
        gts = ",".join(str(float(x)) for x in rng.normal(80, 1, size=len(g)).astype(np.float32))
        hn.append(f"{q}\t{gts}\t{','.join(map(str, fdocs))}\t{','.join(str(float(s)) for s in fs)}")
    for name, lines in (("gt.tsv", gt), ("dense.tsv", dense), ("nci_coarse.tsv", coarse), ("nci_hn.tsv", hn)):
        with open(os.path.join(d, name), "w") as f:
            f.write("\n".join(lines) + "\n")


def g6_consumers():
    """stdout + ofile of the UNMODIFIED reference evaluate.py / ensemble_marco.py on synthetic TSVs."""
    out_dir = os.path.join(GOLD, "g6_consumers")
    shutil.rmtree(out_dir, ignore_errors=True)
    os.makedirs(out_dir)
    synth_consumer_files(out_dir)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    runs = {
        "evaluate_default": ["evaluate.py", "--dir_path", "{d}", "--gt_file", "gt.tsv", "--ance_file", "dense.tsv"],
        "evaluate_recall5_20": ["evaluate.py", "--dir_path", "{d}", "--gt_file", "gt.tsv", "--ance_file", "dense.tsv",
                                "--recall_num", "5,20", "--ofile", "{d}/eval_out.txt"],
        "ensemble_default": ["ensemble_marco.py", "--dir_path", "{d}", "--mapping_file", "{d}/rqmapping.pkl",
                             "--gt_file", "gt.tsv", "--ance_file", "dense.tsv", "--coarse_file", "nci_coarse.tsv",
                             "--fine_file", "nci_hn.tsv", "--ofile", "{d}/ens_out.txt"],
        "ensemble_grid": ["ensemble_marco.py", "--dir_path", "{d}", "--mapping_file", "{d}/rqmapping.pkl",
                          "--gt_file", "gt.tsv", "--ance_file", "dense.tsv", "--coarse_file", "nci_coarse.tsv",
                          "--fine_file", "nci_hn.tsv", "--alphas", "0.3,1.0", "--betas", "0.03,0.5",
                          "--gammas", "0.02,0.5", "--recall_num", "1,10,100"],
        "ensemble_nofine": ["ensemble_marco.py", "--dir_path", "{d}", "--mapping_file", "{d}/rqmapping.pkl",
                            "--gt_file", "gt.tsv", "--ance_file", "dense.tsv", "--coarse_file", "nci_coarse.tsv"],
    }
    expected = {}
    for name, argv in runs.items():
        with tempfile.TemporaryDirectory() as tmp:
            for f in os.listdir(out_dir):
                if f.endswith((".tsv", ".pkl")):
                    shutil.copy(os.path.join(out_dir, f), tmp)
            cmd = [sys.executable] + [a.replace("{d}", tmp) for a in argv]
            cmd[1] = os.path.join(REF, cmd[1])
            r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=tmp)
            assert r.returncode == 0, r.stderr[-2000:]
            ofile = None
            for f in ("eval_out.txt", "ens_out.txt"):
                if os.path.exists(os.path.join(tmp, f)):
                    ofile = open(os.path.join(tmp, f)).read()
            expected[name] = dict(argv=argv, stdout=r.stdout, ofile=ofile)
    json.dump(expected, open(os.path.join(out_dir, "expected.json"), "w"), indent=1)
    print("g6", list(expected))


def g6n_consumers_nq():
    """stdout + ofile of the UNMODIFIED reference ensemble_nqdpr.py on synthetic NQ-style files: the marco TSVs of g6
    plus the inverse answer index (test_inverse_offsets.bin / test_inverse_array.bin: per doc, the questions it answers)."""
    out_dir = os.path.join(GOLD, "g6_consumers_nq")
    shutil.rmtree(out_dir, ignore_errors=True)
    os.makedirs(out_dir)
    synth_consumer_files(out_dir, seed=2)
    rng = np.random.default_rng(3)
    ndoc, nq = 3000, 60
    answers = [[] for _ in range(ndoc)]
    for i, line in enumerate(open(os.path.join(out_dir, "gt.tsv"))):
        for g in line.rstrip("\n").split("\t")[1].split(","):
            answers[int(g)].append(i)
    for d in rng.choice(ndoc, size=400, replace=False):          # other docs that happen to contain an answer
        answers[int(d)].append(int(rng.integers(0, nq)))
    offsets = np.zeros(ndoc + 1, np.int32)
    offsets[1:] = np.cumsum([len(a) for a in answers])
    np.concatenate([np.array(a, np.int32) for a in answers if a]).astype(np.int32).tofile(os.path.join(out_dir, "test_inverse_array.bin"))
    offsets.tofile(os.path.join(out_dir, "test_inverse_offsets.bin"))
    os.remove(os.path.join(out_dir, "gt.tsv"))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    runs = {
        "nq_default": ["ensemble_nqdpr.py", "--dir_path", "{d}", "--mapping_file", "{d}/rqmapping.pkl", "--ance_file", "dense.tsv",
                       "--coarse_file", "nci_coarse.tsv", "--fine_file", "nci_hn.tsv", "--ofile", "{d}/ens_out.txt"],
        "nq_grid_nofine": ["ensemble_nqdpr.py", "--dir_path", "{d}", "--mapping_file", "{d}/rqmapping.pkl", "--ance_file", "dense.tsv",
                           "--coarse_file", "nci_coarse.tsv", "--alphas", "0.3,1.0", "--betas", "0.03,0.5", "--gammas", "0.02,0.5",
                           "--recall_num", "1,10,40"],
        "nq_noensemble": ["ensemble_nqdpr.py", "--dir_path", "{d}", "--mapping_file", "{d}/rqmapping.pkl", "--ance_file", "dense.tsv",
                          "--fine_file", "nci_hn.tsv", "--noensemble"],
    }
    expected = {}
    for name, argv in runs.items():
        with tempfile.TemporaryDirectory() as tmp:
            for f in os.listdir(out_dir):
                if f.endswith((".tsv", ".pkl", ".bin")):
                    shutil.copy(os.path.join(out_dir, f), tmp)
            cmd = [sys.executable] + [a.replace("{d}", tmp) for a in argv]
            cmd[1] = os.path.join(REF, cmd[1])
            r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=tmp)
            assert r.returncode == 0, r.stderr[-2000:]
            ofile = open(os.path.join(tmp, "ens_out.txt")).read() if os.path.exists(os.path.join(tmp, "ens_out.txt")) else None
            expected[name] = dict(argv=argv, stdout=r.stdout, ofile=ofile)
    json.dump(expected, open(os.path.join(out_dir, "expected.json"), "w"), indent=1)
    print("g6n", list(expected))


# --------------------------------------------------------------------------- G7
def g7_writers():
    """Exact bytes of faiss_search.to_file (faiss_search.py:71-77) and LogTxtFile lines
    (main_models.py:244-273) for the coarse / fine / hn tuples infer() logs."""
    ref_import.setup()
    import torch
    import faiss_search
    from main_models import LogTxtFile

    out = {}
    rng = np.random.default_rng(5)
    dists = rng.normal(75, 10, size=(5, 7)).astype(np.float32)
    dists[4, 5:] = -3.4028234663852886e+38
    dists[0, 0] = 100.0
    dists[1, 1] = np.float32(1e-7)
    dists[2, 2] = np.float32(123456.789)
    indices = rng.integers(0, 8841823, size=(5, 7)).astype(np.int64)
    indices[4, 5:] = -1
    with tempfile.TemporaryDirectory() as tmp:
        qf = os.path.join(tmp, "q.tsv")
        with open(qf, "w") as f:
            for i in range(5):
                f.write(f"query text {i} with \"quote\"\t{i},{i + 1}\n")
        of = os.path.join(tmp, "o.tsv")
        faiss_search.to_file(qf, of, dists, indices)
        out["to_file"] = dict(raw_query=open(qf).read(), dists_hex=dists.tobytes().hex(), shape=list(dists.shape),
                              indices=indices.tolist(), expected=open(of).read())
        # LogTxtFile: tuples exactly as infer() builds them
        final = os.path.join(tmp, "res_coarse.tsv")
        for stale in ("/tmp/res_coarse.tsv_0", "/tmp/res_coarse.tsv_1"):
            if os.path.exists(stale):
                os.remove(stale)
        scores32 = torch.tensor([-0.123456789, -1.5, -2.25], dtype=torch.float32)
        lines0 = [("q a", [[1, 2, 3, 4], [5, 6, 7, 8]], [[1, 2, 3, 4]], [float(scores32[0]) / 5 ** 0.8, -0.75])]
        lines1 = [("q b", [3, 1, 2], [9]),
                  ("q c", ",".join(str(s.item()) for s in scores32), "5,6,7", ",".join(str(s.item()) for s in scores32[:2]))]
        l0 = LogTxtFile(final, 0, 2, lambda: None)
        l1 = LogTxtFile(final, 1, 2, lambda: None)
        for ln in lines0:
            l0.add(ln)
        for ln in lines1:
            l1.add(ln)
        l1.wrapped_flush()
        l0.wrapped_merge()
        out["logtxt"] = dict(rank0=json.loads(json.dumps(lines0)), rank1=json.loads(json.dumps(lines1)),
                             expected=open(final).read())
    json.dump(out, open(os.path.join(GOLD, "g7_writers.json"), "w"), indent=1)
    print("g7 ok")

# --------------------------------------------------------------------------- G1 / G2 / G3
def _mevi_t5_config(T5Config, M, K, d_model=32, d_ff=64, heads=4, d_kv=8, layers=2, dec_layers=2,
                    adaptor_layers=2, vocab=512):
    """T5Config exactly as T5FineTuner builds it (main_models.py:1350-1389) for the eval flags of
    marco_eval_nci_rq.sh, shrunk to fixture size."""
    return T5Config(
        vocab_size=vocab, num_layers=layers, num_decoder_layers=dec_layers, d_ff=d_ff, d_model=d_model,
        num_heads=heads, decoder_start_token_id=0, output_past=True, d_kv=d_kv, dropout_rate=0.1,
        decode_embedding=2, hierarchic_decode=0, decode_vocab_size=K * (M + 2) + 2, output_vocab_size=K,
        tie_word_embeddings=0, tie_decode_embedding=1, contrastive=0, Rdrop=0.0, Rdrop_only_decoder=0,
        Rdrop_loss="KL", adaptor_decode=1, adaptor_efficient=1, adaptor_layer_num=adaptor_layers,
        embedding_distillation=0.0, weight_distillation=0.0, input_dropout=1, denoising=0,
        multiple_decoder=0, decoder_num=1, train_batch_size=2, eval_batch_size=2, max_output_length=M + 2,
        use_codebook=1, pq_loss="ce", pq_twin_loss="co", reserve_decoder=0, decoder_integration="series",
        topk_minpooling=None)


def _synthetic_queries(rng, n, seqlen, vocab):
    """token ids as the tokenizer contract produces them: tokens, eos=1, pad=0 (SURVEY 'Tokenizer contract')."""
    ids = np.zeros((n, seqlen), np.int64)
    mask = np.zeros((n, seqlen), np.int64)
    for i in range(n):
        L = int(np.clip(rng.poisson(9) + 2, 3, seqlen)) if i else seqlen   # first query fills the window
        ids[i, :L - 1] = rng.integers(3, vocab, size=L - 1)
        ids[i, L - 1] = 1
        mask[i, :L] = 1
    return ids, mask


def g1_nci_generate():
    """T5ForConditionalGeneration.generate(...) of the vendored MEVI fork with the kwargs infer() passes
    (main_models.py:3612-3641): decoded tokens, scores, step-wise last-position logits, encoder states."""
    ref_import.setup()
    import torch
    from transformers import T5Config, T5ForConditionalGeneration
    from main_models import TreeBuilder, encode_single_newid

    for (M, K, beams, seed) in [(4, 32, 10, 0), (3, 16, 4, 1), (4, 32, 4, 2), (3, 256, 10, 3), (3, 8, 10, 4), (2, 4, 10, 5)]:   # (3, 256) = BASELINE.json configs[2] code shape; the last two: K < R (SURVEY 8a' note ii)
        torch.manual_seed(seed)
        cfg = _mevi_t5_config(T5Config, M, K)
        with io.StringIO() as buf, redirect_stdout(buf):
            model = T5ForConditionalGeneration(cfg)
        model.eval()
        # make the tiny model less degenerate: random (not ones) norm scales, larger relative bias
        with torch.no_grad():
            for n_, p_ in model.named_parameters():
                if n_.endswith("layer_norm.weight") or "final_layer_norm" in n_:
                    p_.copy_(1.0 + 0.2 * torch.randn_like(p_))
                if "relative_attention_bias" in n_:
                    p_.copy_(torch.randn_like(p_))
                if n_.startswith("adaptor.") and n_.endswith("bias"):
                    p_.copy_(0.05 * torch.randn_like(p_))
        args = Namespace(kary=K, position=1, label_length_cutoff=M, max_output_length=M + 2)
        builder = TreeBuilder(share_sons=True)
        newids = [encode_single_newid(args, [i for _ in range(M)]) for i in range(K)]
        for i in range(M):
            builder.add_layer([ids[i] for ids in newids])
        builder.add_layer([1])
        root = builder.build()
        rng = np.random.default_rng(seed + 50)
        ids, mask = _synthetic_queries(rng, 4, 32, cfg.vocab_size)
        step_logits = []
        orig_forward = model.forward

        def spy(*a, **k):
            out = orig_forward(*a, **k)
            step_logits.append(out[0][:, -1, :].detach().clone())
            return out

        model.forward = spy
        kwargs = dict(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), use_cache=False,
                      max_length=M + 2, length_penalty=0.8, num_return_sequences=beams, early_stopping=False,
                      decode_embedding=2, decode_vocab_size=cfg.decode_vocab_size, decode_tree=root,
                      output_hidden_states=True, output_scores=True, decoder_integration="series",
                      decoder_attention_mask=torch.tensor([[1] * (M + 1) + [0]] * 4), num_beams=beams)
        with torch.no_grad():
            outs, scores, enc_h, dec_h = model.generate(**kwargs)
        model.forward = orig_forward
        sd = {k_: v_.detach().numpy() for k_, v_ in model.state_dict().items()}
        np.savez(os.path.join(GOLD, f"g1_nci_M{M}_K{K}_R{beams}.npz"),
                 input_ids=ids, attention_mask=mask, decoded=outs.numpy(), scores=np.array(scores, dtype=np.float64),
                 enc_hidden=enc_h[::beams].numpy(),
                 **{f"step{t}_logits": l.numpy() for t, l in enumerate(step_logits)},
                 **{"w." + k_: v_ for k_, v_ in sd.items()},
                 cfg=np.array(json.dumps(dict(M=M, K=K, beams=beams, d_model=cfg.d_model, d_ff=cfg.d_ff,
                                              num_heads=cfg.num_heads, d_kv=cfg.d_kv, num_layers=cfg.num_layers,
                                              num_decoder_layers=cfg.num_decoder_layers,
                                              adaptor_layer_num=cfg.adaptor_layer_num, vocab_size=cfg.vocab_size,
                                              layer_norm_epsilon=cfg.layer_norm_epsilon,
                                              relative_attention_num_buckets=cfg.relative_attention_num_buckets))))
        print("g1", M, K, beams, "decoded", tuple(outs.shape), "steps", len(step_logits), "score0", scores[0])


def g1t_nci_generate_generic_tree():
    """generate(...) under a GENERIC prefix tree -- TreeBuilder(share_sons=False).add(path) for every existing code path, as
    main_models.build_tree does outside --codebook mode (main_models.py:50-63,1707-1728): a beam may only continue along paths
    of the trie (generation_utils.py:803-818, incl. its "path not in decode tree -> eos" branch).  Cases: a dense trie (every
    beam survives), a sparse one (fewer live candidates than beams at some levels), and one with a single path."""
    ref_import.setup()
    import torch
    from transformers import T5Config, T5ForConditionalGeneration
    from main_models import TreeBuilder, encode_single_newid

    for (M, K, beams, npaths, seed) in [(4, 32, 10, 400, 20), (3, 16, 4, 12, 21), (3, 8, 10, 30, 22), (2, 4, 4, 1, 23)]:
        torch.manual_seed(seed)
        cfg = _mevi_t5_config(T5Config, M, K)
        with io.StringIO() as buf, redirect_stdout(buf):
            model = T5ForConditionalGeneration(cfg)
        model.eval()
        with torch.no_grad():
            for n_, p_ in model.named_parameters():
                if n_.endswith("layer_norm.weight") or "final_layer_norm" in n_:
                    p_.copy_(1.0 + 0.2 * torch.randn_like(p_))
                if "relative_attention_bias" in n_:
                    p_.copy_(torch.randn_like(p_))
                if n_.startswith("adaptor.") and n_.endswith("bias"):
                    p_.copy_(0.05 * torch.randn_like(p_))
        args = Namespace(kary=K, position=1, label_length_cutoff=M, max_output_length=M + 2)
        rng = np.random.default_rng(seed + 50)
        paths = np.unique(rng.integers(0, K, size=(npaths, M)), axis=0)
        builder = TreeBuilder()
        for pth in paths:
            builder.add(encode_single_newid(args, [int(c) for c in pth]))
        root = builder.build()
        ids, mask = _synthetic_queries(rng, 4, 32, cfg.vocab_size)
        kwargs = dict(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), use_cache=False,
                      max_length=M + 2, length_penalty=0.8, num_return_sequences=beams, early_stopping=False,
                      decode_embedding=2, decode_vocab_size=cfg.decode_vocab_size, decode_tree=root,
                      output_hidden_states=True, output_scores=True, decoder_integration="series",
                      decoder_attention_mask=torch.tensor([[1] * (M + 1) + [0]] * 4), num_beams=beams)
        with torch.no_grad():
            outs, scores, enc_h, dec_h = model.generate(**kwargs)
        sd = {k_: v_.detach().numpy() for k_, v_ in model.state_dict().items()}
        np.savez(os.path.join(GOLD, f"g1t_nci_tree_M{M}_K{K}_R{beams}_P{len(paths)}.npz"),
                 input_ids=ids, attention_mask=mask, decoded=outs.numpy(), scores=np.array(scores, dtype=np.float64),
                 paths=paths.astype(np.int32),
                 **{"w." + k_: v_ for k_, v_ in sd.items()},
                 cfg=np.array(json.dumps(dict(M=M, K=K, beams=beams, d_model=cfg.d_model, d_ff=cfg.d_ff,
                                              num_heads=cfg.num_heads, d_kv=cfg.d_kv, num_layers=cfg.num_layers,
                                              num_decoder_layers=cfg.num_decoder_layers,
                                              adaptor_layer_num=cfg.adaptor_layer_num, vocab_size=cfg.vocab_size,
                                              layer_norm_epsilon=cfg.layer_norm_epsilon,
                                              relative_attention_num_buckets=cfg.relative_attention_num_buckets))))
        print("g1t", M, K, beams, "paths", len(paths), "decoded", tuple(outs.shape))
        print(outs.numpy()[:beams], np.round(np.array(scores[:beams]), 4))


def g1a_nci_generate_all():
    """generate(..., eval_all_documents=True, num_beams=1, num_return_sequences=1) -> _generate_all
    (generation_utils.py:507-521,1013-1136): the scores of all K**M code paths per query, as infer() asks for them under
    --use_topic_model 1 --eval_all_documents 1 (main_models.py:3653-3656)."""
    ref_import.setup()
    import torch
    from transformers import T5Config, T5ForConditionalGeneration

    for (M, K, seed) in [(3, 4, 11), (2, 8, 12)]:
        torch.manual_seed(seed)
        cfg = _mevi_t5_config(T5Config, M, K)
        with io.StringIO() as buf, redirect_stdout(buf):
            model = T5ForConditionalGeneration(cfg)
        model.eval()
        with torch.no_grad():
            for n_, p_ in model.named_parameters():
                if n_.endswith("layer_norm.weight") or "final_layer_norm" in n_:
                    p_.copy_(1.0 + 0.2 * torch.randn_like(p_))
                if "relative_attention_bias" in n_:
                    p_.copy_(torch.randn_like(p_))
                if n_.startswith("adaptor.") and n_.endswith("bias"):
                    p_.copy_(0.05 * torch.randn_like(p_))
        rng = np.random.default_rng(seed + 50)
        ids, mask = _synthetic_queries(rng, 3, 32, cfg.vocab_size)
        kwargs = dict(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), use_cache=False,
                      max_length=M + 2, length_penalty=0.8, num_return_sequences=1, early_stopping=False,
                      decode_embedding=2, decode_vocab_size=cfg.decode_vocab_size, decode_tree=None,
                      output_hidden_states=True, output_scores=True, decoder_integration="series",
                      decoder_attention_mask=torch.tensor([[1] * (M + 1) + [0]] * 3), num_beams=1, eval_all_documents=True)
        with torch.no_grad():
            outs, scores, enc_h, _ = model.generate(**kwargs)
        assert outs is None and tuple(scores.shape) == (3, K ** M)
        sd = {k_: v_.detach().numpy() for k_, v_ in model.state_dict().items()}
        np.savez(os.path.join(GOLD, f"g1a_nci_all_M{M}_K{K}.npz"), input_ids=ids, attention_mask=mask,
                 all_scores=scores.numpy(), **{"w." + k_: v_ for k_, v_ in sd.items()},
                 cfg=np.array(json.dumps(dict(M=M, K=K, d_model=cfg.d_model, d_ff=cfg.d_ff,
                                              num_heads=cfg.num_heads, d_kv=cfg.d_kv, num_layers=cfg.num_layers,
                                              num_decoder_layers=cfg.num_decoder_layers,
                                              adaptor_layer_num=cfg.adaptor_layer_num, vocab_size=cfg.vocab_size,
                                              layer_norm_epsilon=cfg.layer_norm_epsilon,
                                              relative_attention_num_buckets=cfg.relative_attention_num_buckets))))
        print("g1a", M, K, "all_scores", tuple(scores.shape), float(scores.max()), float(scores.min()))


def g2_t5_tower():
    """T5Model forward as DocumentEncoder.encode does it (document_encoder.py:104-120): decoder_input_ids = 0,
    reps = last_hidden_state[:, 0, :]; plus encoder/decoder per-layer hidden states."""
    ref_import.setup()
    import torch
    from transformers import T5Config, T5Model

    torch.manual_seed(7)
    cfg = T5Config(vocab_size=512, d_model=32, d_ff=64, num_heads=4, d_kv=8, num_layers=2, num_decoder_layers=2,
                   dropout_rate=0.1)
    model = T5Model(cfg)
    model.eval()
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if n_.endswith("layer_norm.weight"):
                p_.copy_(1.0 + 0.2 * torch.randn_like(p_))
            if "relative_attention_bias" in n_:
                p_.copy_(torch.randn_like(p_))
    rng = np.random.default_rng(77)
    ids, mask = _synthetic_queries(rng, 8, 32, cfg.vocab_size)
    with torch.no_grad():
        out = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask),
                    decoder_input_ids=torch.zeros((8, 1), dtype=torch.long), return_dict=True,
                    output_hidden_states=True)
    np.savez(os.path.join(GOLD, "g2_t5_tower.npz"), input_ids=ids, attention_mask=mask,
             reps=out.last_hidden_state[:, 0, :].numpy(),
             enc_last=out.encoder_last_hidden_state.numpy(),
             **{f"enc_h{i}": h.numpy() for i, h in enumerate(out.encoder_hidden_states)},
             **{f"dec_h{i}": h.numpy() for i, h in enumerate(out.decoder_hidden_states)},
             **{"w." + k_: v_.detach().numpy() for k_, v_ in model.state_dict().items()},
             cfg=np.array(json.dumps(dict(d_model=32, d_ff=64, num_heads=4, d_kv=8, num_layers=2,
                                          num_decoder_layers=2, vocab_size=512, layer_norm_epsilon=cfg.layer_norm_epsilon,
                                          relative_attention_num_buckets=cfg.relative_attention_num_buckets))))
    print("g2 reps", tuple(out.last_hidden_state.shape))


def g2p_t5_passage():
    """The same T5Model forward at the PASSAGE shape of gen_doc_embedding (generate.py:116-187): 128-token windows,
    ragged lengths (one full window, one near-empty), reps = last_hidden_state[:, 0, :]."""
    ref_import.setup()
    import torch
    from transformers import T5Config, T5Model

    torch.manual_seed(7)
    cfg = T5Config(vocab_size=512, d_model=32, d_ff=64, num_heads=4, d_kv=8, num_layers=2, num_decoder_layers=2,
                   dropout_rate=0.1)
    model = T5Model(cfg)
    model.eval()
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if n_.endswith("layer_norm.weight"):
                p_.copy_(1.0 + 0.2 * torch.randn_like(p_))
            if "relative_attention_bias" in n_:
                p_.copy_(torch.randn_like(p_))
    rng = np.random.default_rng(78)
    n, S = 6, 128
    ids = np.zeros((n, S), np.int64)
    mask = np.zeros((n, S), np.int64)
    for i, L in enumerate([128, 3, 65, 64, 100, 37]):
        ids[i, :L - 1] = rng.integers(3, cfg.vocab_size, size=L - 1)
        ids[i, L - 1] = 1
        mask[i, :L] = 1
    with torch.no_grad():
        out = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask),
                    decoder_input_ids=torch.zeros((n, 1), dtype=torch.long), return_dict=True)
    np.savez(os.path.join(GOLD, "g2p_t5_passage.npz"), input_ids=ids, attention_mask=mask,
             reps=out.last_hidden_state[:, 0, :].numpy(), enc_last=out.encoder_last_hidden_state.numpy(),
             **{"w." + k_: v_.detach().numpy() for k_, v_ in model.state_dict().items()},
             cfg=np.array(json.dumps(dict(d_model=32, d_ff=64, num_heads=4, d_kv=8, num_layers=2,
                                          num_decoder_layers=2, vocab_size=512, layer_norm_epsilon=cfg.layer_norm_epsilon,
                                          relative_attention_num_buckets=cfg.relative_attention_num_buckets))))
    print("g2p reps", tuple(out.last_hidden_state.shape))


def g8_bert_tower():
    """BertModel forward as DocumentEncoder.encode does it for mtype 'bert' (document_encoder.py:104-120):
    reps = last_hidden_state[:, 0, :]; 40-token windows with ragged lengths, token types all 0."""
    ref_import.setup()
    import torch
    from transformers import BertConfig, BertModel

    torch.manual_seed(9)
    cfg = BertConfig(vocab_size=400, hidden_size=48, num_hidden_layers=2, num_attention_heads=4, intermediate_size=96,
                     max_position_embeddings=64, type_vocab_size=2)
    model = BertModel(cfg)
    model.eval()
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if n_.endswith("LayerNorm.weight"):
                p_.copy_(1.0 + 0.2 * torch.randn_like(p_))
            elif n_.endswith("bias"):
                p_.copy_(0.1 * torch.randn_like(p_))
            elif "embeddings" not in n_:
                p_.copy_(0.15 * torch.randn_like(p_))
    rng = np.random.default_rng(80)
    n, S = 7, 40
    ids = np.zeros((n, S), np.int64)
    mask = np.zeros((n, S), np.int64)
    for i, L in enumerate([40, 3, 17, 33, 8, 25, 40]):
        ids[i, 0] = 101
        ids[i, 1:L - 1] = rng.integers(5, cfg.vocab_size, size=L - 2)
        ids[i, L - 1] = 102
        mask[i, :L] = 1
    with torch.no_grad():
        out = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask),
                    token_type_ids=torch.zeros((n, S), dtype=torch.long), return_dict=True)
    sd = {k_: v_ for k_, v_ in model.state_dict().items() if "position_ids" not in k_ and not k_.startswith("pooler")}
    np.savez(os.path.join(GOLD, "g8_bert_tower.npz"), input_ids=ids, attention_mask=mask,
             hidden=out.last_hidden_state.numpy(), reps=out.last_hidden_state[:, 0, :].numpy(),
             **{"w." + k_: v_.detach().numpy() for k_, v_ in sd.items()},
             cfg=np.array(json.dumps(dict(num_hidden_layers=2, num_attention_heads=4, layer_norm_eps=cfg.layer_norm_eps,
                                          hidden_act=cfg.hidden_act))))
    print("g8 reps", tuple(out.last_hidden_state.shape), cfg.hidden_act)


def g3_relative_buckets():
    """T5Attention._relative_position_bucket tables (modeling_t5.py:241-304)."""
    ref_import.setup()
    import torch
    from transformers.modeling_t5 import T5Attention

    out = {}
    for name, (ql, kl, bidir) in dict(enc32=(32, 32, True), dec6=(6, 6, False), enc200=(200, 200, True),
                                      dec150=(150, 150, False)).items():
        ctx = torch.arange(ql)[:, None]
        mem = torch.arange(kl)[None, :]
        out[name] = T5Attention._relative_position_bucket(mem - ctx, bidirectional=bidir, num_buckets=32).numpy()
    np.savez_compressed(os.path.join(GOLD, "g3_relative_buckets.npz"), **out)
    print("g3 ok")


# --------------------------------------------------------------------------- G9
def g9_ip_rank():
    """Inner-product ranking as the reference itself executes it (pins the dense / fine-stage oracle):

    (i)  fine stage of infer(), main_models.py:3921-4013: per beam cluster `doc_cluster.get(code)`, rows gathered
         from the embedding matrix, `DocumentEncoder.generate(q, p_reps=rows).scores` (= compute_similarity =
         torch.matmul, document_encoder.py:128-132,213-226) in encode_batch_size slices, torch.cat, np.concatenate,
         `torch.sort(scores, descending=True)`; plus the gt-document scores of the hard-negative line (:4024-4045);
    (ii) the --eval_all_documents streaming loop, main_models.py:3818-3876: chunked generate() scores,
         get_inference_scores(..., eval_all=True) (:3539-3552), running `torch.topk` over cat(stack, new).

    T5FineTunerWithValidation cannot be constructed here (needs spiece.model), so the driver lines are re-issued
    around the reference's own DocumentEncoder / get_inference_scores objects and the very torch calls infer() makes.
    Two data sets: 'int' -- small-integer-valued f32 (every partial sum is exact, so BLAS order and the sequential
    fmaf chain give the same bits; many exact ties), 'flt' -- anisotropic gaussian embeddings (no ties in practice;
    compared within f32 summation-order tolerance)."""
    ref_import.setup()
    import torch
    from transformers import T5Config, T5Model
    from document_encoder import DocumentEncoder
    import main_models

    torch.manual_seed(9)
    tiny = T5Model(T5Config(vocab_size=32, d_model=8, d_ff=8, num_heads=2, d_kv=4, num_layers=1, num_decoder_layers=1))
    denc = DocumentEncoder(lm_q=tiny, lm_p=tiny)
    denc.eval()
    fake_self = Namespace(args=Namespace(use_topic_model=0))
    get_inference_scores = main_models.T5FineTunerWithValidation.get_inference_scores

    for name, N, dim, M, K, R, nq, ebs, pools in [("int", 1200, 768, 3, 5, 6, 12, 7, (50, 2000)),
                                                   ("flt", 1500, 128, 3, 6, 10, 16, 64, (100, 1000))]:
        rng = np.random.default_rng(900 + dim)
        if name == "int":
            emb = rng.integers(-6, 7, size=(N, dim)).astype(np.float32)
            q = rng.integers(-6, 7, size=(nq, dim)).astype(np.float32)
            emb[100:110] = emb[5]                      # duplicated passages (exact ties over different ids)
            emb[700] = emb[699]
        else:
            mean = rng.standard_normal(dim).astype(np.float32)
            emb = (mean + 0.3 * rng.standard_normal((N, dim))).astype(np.float32)
            q = (mean + 0.3 * rng.standard_normal((nq, dim))).astype(np.float32)
            q[:8] = emb[rng.integers(0, N, size=8)] + 0.05 * rng.standard_normal((8, dim)).astype(np.float32)
        codes = rng.integers(0, K, size=(N, M)).astype(np.int32)
        codes[codes[:, 0] == K - 1, 0] = 0             # no document has a first code of K-1: those clusters are empty
        doc_cluster = {}
        for i, c in enumerate(codes):
            doc_cluster.setdefault(tuple(int(x) for x in c), []).append(i)
        beams = rng.integers(0, K, size=(nq, R, M)).astype(np.int64)
        beams[0] = codes[rng.integers(0, N, size=R)]
        beams[1, 2] = beams[1, 0]                      # a cluster repeated in one beam list (scored twice by infer())
        beams[2, :, 0] = K - 1                         # every beam cluster empty
        beams[3, 1, 0] = K - 1                         # one empty cluster among populated ones
        gt = [[int(x) for x in rng.integers(0, N, size=1 + (i % 2))] for i in range(nq)]

        all_embeddings = torch.from_numpy(emb)
        query_embedding = torch.from_numpy(q)
        # ---- (i) fine stage: main_models.py:3813-3815 expands the query embedding once per beam
        qe = query_embedding.unsqueeze(1).expand(-1, R, -1).reshape(-1, dim)
        out_docs, out_scores, out_gt, seg, ndoc_all = [], [], [], [0], []
        q_ind = 0
        with torch.no_grad():
            for didx in range(nq):
                cur_ndoc, scores, docs = 0, [], []
                for i in range(R):
                    cur_docs = doc_cluster.get(tuple(int(x) for x in beams[didx, i]), None)
                    if cur_docs is not None:
                        cur_ndoc += len(cur_docs)
                        doc_embedding = all_embeddings[cur_docs]
                        for start in range(0, len(doc_embedding), ebs):
                            output = denc.generate(qe[q_ind], p_reps=doc_embedding[start:start + ebs])
                            scores.append(get_inference_scores(fake_self, None, 0, output.scores))
                        docs.append(cur_docs)
                    q_ind += 1
                if len(scores) > 0:
                    scores = torch.cat(scores)
                    docs = np.concatenate(docs, dtype=int)
                    scores, index = torch.sort(scores, descending=True)
                    sorted_docs = docs[index.cpu().numpy()].tolist()
                    scores = scores.numpy()
                else:
                    sorted_docs, scores = [], np.zeros(0, np.float32)
                out_docs += sorted_docs
                out_scores.append(np.asarray(scores, np.float32))
                seg.append(len(out_docs))
                ndoc_all.append(cur_ndoc)
                gt_out = denc.generate(qe[q_ind - R], p_reps=all_embeddings[gt[didx]])
                out_gt.append(gt_out.scores.numpy().astype(np.float32))
        res = dict(emb=emb, q=q, codes=codes, beams=beams, M=M, K=K, encode_batch_size=ebs,
                   gt_flat=np.array([d for g in gt for d in g], np.int64), gt_len=np.array([len(g) for g in gt], np.int64),
                   fine_docs=np.array(out_docs, np.int64), fine_scores=np.concatenate(out_scores), fine_seg=np.array(seg, np.int64),
                   fine_ndoc=np.array(ndoc_all, np.int64), gt_scores=np.concatenate(out_gt))
        # ---- (ii) eval_all_documents streaming top-k
        for pool_size in pools:
            with torch.no_grad():
                stack_scores = torch.empty([nq, 0], dtype=torch.float)
                sorted_docs = torch.empty([nq, 0], dtype=torch.int32)
                for start in range(0, N, ebs * 16):
                    ending = min(start + ebs * 16, N)
                    output = denc.generate(query_embedding, p_reps=all_embeddings[start:ending], bmm=False)
                    new_scores = get_inference_scores(fake_self, None, 0, output.scores, eval_all=True)
                    new_docs = torch.arange(start, ending, dtype=sorted_docs.dtype).unsqueeze(0).expand(nq, -1)
                    scores = torch.cat([stack_scores, new_scores], dim=-1)
                    docs = torch.cat([sorted_docs, new_docs], dim=-1)
                    stack_scores, doc_indices = torch.topk(scores, k=min(scores.shape[-1], pool_size), dim=-1)
                    sorted_docs = docs.gather(-1, doc_indices)
            res[f"all{pool_size}_docs"] = sorted_docs.numpy().astype(np.int64)
            res[f"all{pool_size}_scores"] = stack_scores.numpy()
            res[f"all{pool_size}_last_scores"] = scores.numpy()[:, :8].copy()   # the hn line's `scores` quirk (:3904-3907)
        np.savez_compressed(os.path.join(GOLD, f"g9_ip_rank_{name}.npz"), **res)
        print("g9", name, "fine lists", len(out_docs), "ndoc", ndoc_all[:6])


ALL = dict(g4=g4_rq, g5=g5_tree_codec, g6=g6_consumers, g6n=g6n_consumers_nq, g7=g7_writers, g1=g1_nci_generate, g2=g2_t5_tower, g2p=g2p_t5_passage, g8=g8_bert_tower,
           g3=g3_relative_buckets, g9=g9_ip_rank, g1a=g1a_nci_generate_all, g1t=g1t_nci_generate_generic_tree)

if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or list(ALL)
    for w in which:
        ALL[w]()
