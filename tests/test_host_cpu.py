"""CPU tests of host logic: CLI argv surfaces, rank partitioning rules, token codec / tree goldens (G5),
metric aggregation, and the sharded dense arm's collective plumbing on gloo (world_size 2)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

EVAL_ARGV = """--n_gpu 8 --mode eval --query_type gtq --co_neg_from clus --model_info base --id_class bert_k30_c30_1
--dataset marco --Rdrop 0. --eval_batch_size 2 --encode_batch_size 1024 --document_encoder ance --recall_level both
--qtower encmask_dec --query_embed_accum attenpool --pq_loss ce --pq_type rq --codebook 1 --subvector_num 4
--subvector_bits 5 --use_gumbel_softmax 0 --pq_softmax_tau 1 --pq_hard_softmax_topk 1 --pq_negative none --fixnci
--fixpq --document_encoder_from_pretrained 0 --not_load_document_encoder 0 --no_nci_loss 1 --query_encoder twin
--num_return_sequences 10 --save_hard_neg 8841823 --pq_path D/ance/rqcodebook4_5.pt --pq_cluster_path D/ance/rqclus4_5.pkl
--nci_ckpt D/ckpts/nci.ckpt --data_dir D/origin --newid_dir D/ance --document_path D/ance/all_document --ckpt_dir D/ckpts
--embedding_path D/ance/docemb.bin --custom_save_path D/ance/nci_result_rq45_top10.tsv""".split()


def test_main_accepts_every_flag_of_marco_eval_nci_rq_sh():
    import main

    a = main.parsers_parser(EVAL_ARGV)
    main.check_supported(a)
    assert a.n_gpu == list(range(8)) and a.num_return_sequences == 10 and a.save_hard_neg == 8841823
    assert a.subvector_num == 4 and a.subvector_bits == 5 and a.length_penalty == 0.8
    assert ("--fixnci", None) in a.ignored_flags and ("--Rdrop", "0.") in a.ignored_flags
    assert main.parsers_parser(["--mode", "eval", "--data_dir", "x", "--n_gpu", "[2,5]"]).n_gpu == [2, 5]
    with pytest.raises(SystemExit):
        main.check_supported(main.parsers_parser(["--mode", "train", "--data_dir", "x"]))
    # marco_generate_embedding_n_rq.sh: the offline index build is the one --mode train configuration that is built
    gen = """--n_gpu 8 --mode train --query_type gtq --co_neg_from clus --model_info base --id_class bert_k30_c30_1 --dataset marco
    --encode_batch_size 1024 --recall_level fine --pq_type rq --only_gen_rq 1 --codebook 1 --subvector_num 4 --subvector_bits 5
    --document_encoder_from_pretrained 0 --not_load_document_encoder 0 --no_nci_loss 1 --query_encoder twin
    --num_return_sequences 10 --document_encoder ance --pq_path D/ance/rqcodebook4_5.pt --pq_cluster_path D/ance/rqclus4_5.pkl
    --data_dir D/origin --ckpt_dir D/ckpts --newid_dir D/ance --document_path D/ance/all_document
    --embedding_path D/ance/docemb.bin""".split()
    g = main.parsers_parser(gen)
    main.check_supported(g)
    assert g.only_gen_rq == 1 and g.co_doc_length == 128 and g.seed == 42
    with pytest.raises(SystemExit):
        i = gen.index("--only_gen_rq")
        main.check_supported(main.parsers_parser(gen[:i] + gen[i + 2:]))        # plain training is not built
    main.check_supported(main.parsers_parser(EVAL_ARGV + ["--document_encoder", "ar2"]))   # BERT-family towers are built
    with pytest.raises(SystemExit):
        main.check_supported(main.parsers_parser(EVAL_ARGV + ["--document_encoder", "dpr"]))
    with pytest.raises(SystemExit):   # the brute-force ablation needs recall_level fine + knn_topk_by_step 1 (MEVI/main.py:657-658)
        main.check_supported(main.parsers_parser(EVAL_ARGV + ["--eval_all_documents", "1"]))
    main.check_supported(main.parsers_parser(EVAL_ARGV + ["--eval_all_documents", "1", "--recall_level", "fine",
                                                          "--knn_topk_by_step", "1"]))
    main.check_supported(main.parsers_parser(EVAL_ARGV + ["--doc_multiclus", "3", "--multiclus_score_aggr", "max"]))
    with pytest.raises(SystemExit):   # other ablation modes are not built
        main.check_supported(main.parsers_parser(EVAL_ARGV + ["--pq_type", "opq"]))


def test_cli_argument_surfaces_match_reference():
    for script, flags in {
        "faiss_search.py": ["--query_path", "--doc_path", "--output_path", "--raw_query_path", "--dim", "--topk", "--param"],
        "generate.py": ["--gen_doc", "--document_dir", "--doc_embedding_path", "--query_file", "--model_path", "--tokenizer_path", "--query_embedding_path", "--ckpt_path",
                        "--batch_size", "--dim", "--gpus", "--gen_query", "--timing_infer_step"],
        "evaluate.py": ["--dir_path", "--gt_file", "--ance_file", "--recall_num", "--ofile"],
        "ensemble_marco.py": ["--dir_path", "--gt_file", "--ance_file", "--fine_file", "--coarse_file", "--mapping_file",
                              "--alphas", "--betas", "--gammas", "--recall_num", "--ofile"],
    }.items():
        r = subprocess.run([sys.executable, os.path.join(ROOT, script), "--help"], capture_output=True, text=True,
                           env=dict(os.environ, PYTHONPATH=ROOT))
        assert r.returncode == 0, r.stderr[-500:]
        for f in flags:
            assert f in r.stdout, (script, f)


def test_rank_partition_rules():
    import generate
    from mevi_amd.dense import shard_range
    from mevi_amd.evalrun import rank_slice

    # generate.py:74-82 -- the first n % nrank ranks take one more row, ranges are contiguous
    for n, w in [(6980, 8), (10, 3), (5, 8)]:
        r = [generate.rank_range(n, i, w) for i in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        sizes = [b - a for a, b in r]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    # documents (generate.py:141-147): n // nrank each, the last rank takes the remainder
    assert [generate.doc_rank_range(10, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 10)]
    assert generate.doc_rank_range(8841823, 7, 8) == (7 * 1105227, 8841823)
    # dense arm: ceil(N / W) rows per rank
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert shard_range(8841823, 7, 8) == (7 * 1105228, 8841823)
    # DistributedSampler(shuffle=False): strided, padded by repeating the head
    assert rank_slice(10, 0, 4) == [0, 4, 8] and rank_slice(10, 3, 4) == [3, 7, 1]
    from torch.utils.data import DistributedSampler

    for n, w in [(10, 4), (23, 8), (7, 2)]:
        for r in range(w):
            assert rank_slice(n, r, w) == list(DistributedSampler(range(n), num_replicas=w, rank=r, shuffle=False))


def test_token_codec_against_reference_golden():
    from mevi_amd.nci import dec_2d, decode_token

    for case in json.load(open(os.path.join(GOLD, "g5_tree_codec.json"))):
        M, K = case["M"], case["K"]
        enc = torch.tensor(case["encoded"])
        assert enc.tolist() == [[2 + p * K + c for p, c in enumerate(row)] + [1] for row in case["codes"]]
        assert case["encoded_from_str"] == enc[0].tolist()
        assert case["tree_levels"] == [[2 + p * K + c for c in range(K)] for p in range(M)] + [[1]]
        seqs = torch.cat([torch.zeros((len(enc), 1), dtype=torch.long), enc], 1)
        got = decode_token(seqs, K)
        assert got.tolist() == case["decoded"] == case["codes"] and case["eos_is_none"]
        assert dec_2d(got, 3).tolist() == case["dec_2d"]
        assert dec_2d(list(range(7)), 3) == [[0, 1, 2], [3, 4, 5], [6]]


def test_metric_aggregation_matches_consumer_semantics():
    from mevi_amd.evalrun import summarize

    res = [("q1", 12, (0, None), (3, None)), ("q2", 0, (None,), (None,)), ("q3", 5, (7,), (0,)),
           ("q1", 12, (0, None), (3, None))]            # DistributedSampler padding duplicates q1
    out = summarize(res, [1, 5, 10, 20], 10)
    assert out["nqueries"] == 3 and abs(out["ndoc"] - 17 / 3) < 1e-12
    assert out["recall"][5] == (0.5 + 0 + 1) / 3 and out["mrr"][5] == (1 / 4 + 0 + 1) / 3 and out["recall"][1] == 1 / 3
    assert out["cluster_recall"][10] == (0.5 + 0 + 1) / 3 and out["cluster_recall"][5] == 0.5 / 3
    assert list(out["cluster_recall"]) == [1, 5, 10]
    # recall_level 'fine' (--eval_all_documents): 3-tuples, the extra 'cluster<R>' key = found-at-all figures
    fine = summarize([("q1", 50, (3, None)), ("q2", 50, (None,)), ("q3", 50, (0,))], [1, 5], 10, both=False)
    assert "cluster_recall" not in fine and fine["ndoc"] == 50
    assert fine["recall"] == {1: 1 / 3, 5: (0.5 + 0 + 1) / 3, "cluster10": (0.5 + 0 + 1) / 3}
    assert fine["mrr"]["cluster10"] == (1 / 4 + 0 + 1) / 3 and fine["hitrate"]["cluster10"] == 2 / 3


def _gloo_worker(rank, world, port, q, d, k, ret):
    import torch.distributed as dist

    from mevi_amd import dense
    from oracle import dense as od

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = dense.shard_range(len(d), rank, world)

    rounds = []

    def local(qq, dd, kk, id_offset=0):     # CPU stand-ins for the two device steps (checker = the oracle)
        rounds.append((qq.shape[0], kk))
        s, i = od.ip_topk_exact(qq.numpy(), dd.numpy(), kk, id_offset)
        return torch.from_numpy(s), torch.from_numpy(i)

    def merge(s, i, kk):
        ms, mi = od.topk_merge(s.numpy(), i.numpy(), kk)
        return torch.from_numpy(ms), torch.from_numpy(mi)

    s, i = dense.sharded_ip_topk(torch.from_numpy(q), torch.from_numpy(d[a:b]), k, id_offset=a,
                                 local_search=local, merge=merge)
    ret[rank] = (s.numpy(), i.numpy(), rounds)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_dense_exchange_on_gloo_world2():
    """The N > 1 path of the dense arm: shard ranges, global ids, all-gather of per-shard top-k and the
    merge must give every rank the un-sharded answer.  gloo / CPU: the device kernels are replaced by the
    oracle through the injection points of sharded_ip_topk, the collective plumbing is the product's."""
    import socket

    import torch.multiprocessing as mp

    from oracle import dense as od

    rng = np.random.default_rng(3)
    q = rng.standard_normal((9, 32)).astype(np.float32)
    d = rng.standard_normal((2001, 32)).astype(np.float32)
    d[:400] += 1.5 * q[0]     # query 0's top-300 sits in rank 0's shard: round 1 (k_local < k) must be redone
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_gloo_worker, nprocs=2, args=(2, port, q, d, 20, ret))   # k_local = k: single round
    es, ei = od.ip_topk_exact(q, d, 20)
    for r in range(2):
        assert np.array_equal(ret[r][1], ei) and np.array_equal(ret[r][0], es)
    import mevi_amd.dense as md
    assert md.truncated_list_len(20, 2) == 20 and md.truncated_list_len(300, 2) == 150 + 5 * 13 + 8
    mp.spawn(_gloo_worker, nprocs=2, args=(2, port + 1, q, d, 300, ret))  # truncated round + redo for query 0
    es, ei = od.ip_topk_exact(q, d, 300)
    for r in range(2):
        assert np.array_equal(ret[r][1], ei) and np.array_equal(ret[r][0], es)
        rounds = ret[r][2]                           # all queries truncated, then the skewed few with full lists
        assert rounds[0] == (9, 223) and len(rounds) == 2 and rounds[1][1] == 300 and 1 <= rounds[1][0] <= 6


def test_nq_answer_index_matches_the_reference_hit_test(tmp_path):
    """NqAnswers (question -> docs, one np.isin per list) against the reference's per-document membership loop
    (metrics.nq_first_hit = ensemble_nqdpr.py:27-31 / main_models.py:4064-4069), incl. padded -1 ids and no-hit lists."""
    from mevi_amd.evalrun import NqAnswers, load_nq_queries
    from mevi_amd.metrics import nq_first_hit

    rng = np.random.default_rng(11)
    ndocs, nquestions = 500, 40
    lists = [sorted(rng.choice(nquestions, size=rng.integers(0, 4), replace=False).tolist()) for _ in range(ndocs)]
    offsets = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.int32)
    array = np.array([q for l in lists for q in l], dtype=np.int32)
    offsets.tofile(tmp_path / "test_inverse_offsets.bin")
    array.tofile(tmp_path / "test_inverse_array.bin")
    with open(tmp_path / "nq-test.qa.csv", "w") as f:
        for i in range(nquestions):
            f.write(f"question {i}\t['answer {i}']\n")
    df = load_nq_queries(str(tmp_path))
    assert df["oldid"].tolist() == list(range(nquestions)) and df["query"][3] == "question 3"
    nq = NqAnswers(str(tmp_path))
    for qind in range(nquestions):
        assert sorted(nq.docs_answering(qind).tolist()) == [d for d, l in enumerate(lists) if qind in l]
        for _ in range(5):
            ranked = rng.choice(ndocs, size=rng.integers(0, 60), replace=False).tolist()
            if ranked and rng.random() < 0.3:
                ranked[rng.integers(0, len(ranked))] = -1
            assert nq.first_hit(qind, ranked) == nq_first_hit(qind, ranked, offsets, array)


def test_native_number_formatting_equals_python_str():
    """textio.hip (host code of libmevi_hip.so) must write what `str(float(x))` / `str(int)` write: random f32 bit
    patterns (subnormals, huge, tiny), the notation switch points, specials."""
    from mevi_amd.io import join_f32, join_i64

    rng = np.random.default_rng(0)
    with np.errstate(invalid="ignore"):
        rand = rng.integers(0, 2 ** 32, size=50000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    special = np.array([0.0, -0.0, 1.0, -1.0, 0.1, 1e-4, 9.999e-5, 1e-5, 1e16, 9.9999999e15, 1.5e16, 123456.0, 1e22, 3.4e38,
                        1.4e-45, np.inf, -np.inf, np.nan, 100.0, 612.34564, 0.5, 2.0 ** 24, 2.0 ** 53], dtype=np.float32)
    for arr in (special, rand):
        with np.errstate(invalid="ignore"):          # signalling-NaN bit patterns in `rand`
            want = [repr(x) for x in arr.astype(np.float64).tolist()]
        assert join_f32(arr).split(",") == want
    ids = np.array([0, -1, 8841822, 2 ** 40, -2 ** 62, 2 ** 63 - 1], dtype=np.int64)
    assert join_i64(ids) == ",".join(map(str, ids.tolist())) and join_i64(ids[:0]) == "" and join_f32(special[:0]) == ""
    # and back (the consumers' field parser): same lists as int() / float() on every token, None for anything else
    from mevi_amd.io import _native_numbers, parse_list

    finite = rand[np.isfinite(rand)]
    text = join_f32(finite)
    assert _native_numbers(text) == [float(x) for x in text.split(",")] == parse_list(text)
    many = rng.integers(-5, 8841823, size=5000)
    assert parse_list(join_i64(many)) == many.tolist() and isinstance(parse_list(join_i64(many))[0], int)
    for odd in ["1,2,x", "1_000,2", "0x10,2", "1,,2", "1,2,", "--1", ""]:
        assert _native_numbers(odd) is None if odd else True
    assert _native_numbers(" 3, 4 ,5") == [3, 4, 5] and _native_numbers("+3,-4") == [3, -4]
    assert str(_native_numbers("inf,-inf,nan,1e5")) == str([float("inf"), float("-inf"), float("nan"), 1e5])


def _host_flow_worker(rank, world, port, tmp, ret):
    import torch.distributed as dist

    from mevi_amd import io as mio
    from mevi_amd.evalrun import rank_slice
    from mevi_amd.indexbuild import embed_documents

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # (1) the per-rank result logs of main.py: every rank appends its DistributedSampler slice, rank 0 concatenates
    log = mio.RankLog(os.path.join(tmp, "out_fine.tsv"), rank, world, barrier=dist.barrier, tmpdir=tmp)
    for i in rank_slice(7, rank, world):
        log.add((f"query {i}", [i, i + 1], [i]))
    log.merge()
    # (2) the per-rank part files of the passage embeddings (main.py --only_gen_rq / generate.py --gen_doc)

    class Enc:
        dim = 3

        def encode_passage(self, psg):
            t = psg["input_ids"].float()
            return torch.stack([t.sum(1), t[:, 0], psg["attention_mask"].sum(1).float()], 1)

    tokens = np.arange(11 * 4, dtype=np.int64).reshape(11, 4)
    masks = (tokens % 3 != 0).astype(np.int64)
    embed_documents(Enc(), tokens, masks, os.path.join(tmp, "docemb.bin"), rank, world, dist.barrier, batch_size=2)
    ret[rank] = True
    dist.barrier()
    dist.destroy_process_group()


def test_multi_rank_host_flows_on_gloo_world2(tmp_path):
    """What main.py / generate.py do around the kernels with N > 1 ranks, with real processes and a gloo barrier: the
    rank-merged log file keeps rank order with the sampler's padding duplicate, the passage-embedding part files are
    concatenated in document order and removed."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_host_flow_worker, nprocs=2, args=(2, port, str(tmp_path), ret))
    assert ret[0] and ret[1]
    lines = open(tmp_path / "out_fine.tsv").read().splitlines()
    # DistributedSampler(shuffle=False) over 7 samples, 2 ranks: rank 0 -> 0,2,4,6; rank 1 -> 1,3,5,0 (padding repeats the head)
    assert [l.split("\t")[0] for l in lines] == [f"query {i}" for i in (0, 2, 4, 6, 1, 3, 5, 0)]
    assert lines[1] == "query 2\t[2, 3]\t[2]"
    emb = np.fromfile(tmp_path / "docemb.bin", dtype=np.float32).reshape(11, 3)
    tokens = np.arange(44, dtype=np.int64).reshape(11, 4)
    want = np.stack([tokens.sum(1), tokens[:, 0], (tokens % 3 != 0).sum(1)], 1).astype(np.float32)
    assert np.array_equal(emb, want)
    assert sorted(os.listdir(tmp_path)) == ["docemb.bin", "out_fine.tsv"]


def test_main_py_rejects_what_it_does_not_build():
    """Only flags of the reference's own parser are accepted (a typo ends the run as argparse would); ablation flags that
    change what --mode eval computes are refused unless they carry the built value."""
    import main

    base = ["--mode", "eval", "--data_dir", "x"]
    a = main.parsers_parser(base + ["--learning_rate", "2e-4", "--fixnci", "--simans_hyper_b", "-1", "--Rdrop=0.1"])
    assert ("--learning_rate", "2e-4") in a.ignored_flags and ("--simans_hyper_b", "-1") in a.ignored_flags
    assert ("--Rdrop", "0.1") in a.ignored_flags
    for argv in (["--lerning_rate", "1"], ["stray"], ["--learning_rate"], ["--fp_16", "1"],
                 ["--cat_cluster_centroid", "2"], ["--cluster_position_topk", "5"], ["--decode_embedding", "1"],
                 ["--infer_reconstruct_vector", "1"], ["--load_encoder_only", "1"], ["--drop_data_rate", "0.1"]):
        with pytest.raises(SystemExit):
            main.parsers_parser(base + argv)
    main.parsers_parser(base + ["--drop_data_rate", "0"])
    with pytest.raises(SystemExit):            # load_data_infer only knows the dev set (main_utils.py:238)
        main.check_supported(main.parsers_parser(EVAL_ARGV + ["--test_set", "test"]))
    main.parsers_parser(base + ["--use_topic_model", "0", "--fp_16", "0", "--decode_embedding", "2"])   # the built values
    main.check_supported(main.parsers_parser(EVAL_ARGV + ["--use_topic_model", "1"]))                  # cluster score x q.d
    main.check_supported(main.parsers_parser(EVAL_ARGV + ["--use_topic_model", "1", "--topic_score_ratio", "0.3"]))
    # the topic model over multi-cluster documents: built on the cluster path (round 3), not with --eval_all_documents
    main.check_supported(main.parsers_parser(EVAL_ARGV + ["--use_topic_model", "1", "--doc_multiclus", "3"]))
    with pytest.raises(SystemExit):
        main.check_supported(main.parsers_parser(EVAL_ARGV + ["--use_topic_model", "1", "--doc_multiclus", "3", "--eval_all_documents", "1",
                                                              "--recall_level", "fine", "--knn_topk_by_step", "1"]))
    with pytest.raises(SystemExit):            # try_load_ckpt asserts a checkpoint (MEVI/main.py:201)
        main.check_supported(main.parsers_parser([t for t in EVAL_ARGV if not t.startswith("--nci_ckpt")][:0] + base + [
            "--codebook", "1", "--pq_type", "rq", "--query_encoder", "twin", "--recall_level", "both", "--document_encoder",
            "ance", "--pq_path", "p", "--pq_cluster_path", "c", "--embedding_path", "e", "--custom_save_path", "o.tsv"]))


def test_checkpoint_loading_rules():
    """try_load_ckpt (MEVI/main.py:198-248): the NCI branch strips 'model.', reports foreign keys and shape mismatches as
    `Bad parameter <k>.`; the whole-model branch (--infer_ckpt) routes model. / document_encoder.lm_q. / pq.codebook and
    drops the four filtered relative_attention_bias keys."""
    import torch

    from mevi_amd import nci
    from mevi_amd.evalrun import nci_weights_from_state_dict, split_whole_checkpoint

    M, K = 3, 4
    cfg = nci.NCIConfig(M=M, K=K, adaptor_layer_num=1, num_decoder_layers=1, num_layers=1, d_model=8, d_ff=16, num_heads=2, d_kv=4)
    good = {n: torch.zeros([s if s is not None else 5 for s in shape]) for n, shape in nci.expected_shapes(cfg).items()}
    said = []
    sd = {"model." + k: v for k, v in good.items()}
    sd.update({"document_encoder.lm_q.shared.weight": torch.zeros(3, 8), "pq.codebook": torch.zeros(M, K, 8), "epoch": 3})
    w = nci_weights_from_state_dict(sd, said.append)
    assert sorted(said) == ["Bad parameter document_encoder.lm_q.shared.weight.", "Bad parameter pq.codebook."]
    assert set(w) == set(good) and nci.check_weights(w, cfg, said.append) == []
    w["decode_embeddings.weight"] = torch.zeros(cfg.V + 4, 8)             # a checkpoint trained with another codebook size
    del w["decoder.final_layer_norm.weight"]
    said.clear()
    assert nci.check_weights(w, cfg, said.append) == ["decode_embeddings.weight", "decoder.final_layer_norm.weight"]
    assert said == ["Bad parameter decode_embeddings.weight.", "Bad parameter decoder.final_layer_norm.weight."]
    bad_key = "model.decoder.block.0.layer.1.EncDecAttention.relative_attention_bias.weight"
    sd[bad_key] = torch.zeros(32, 2)
    nw, tower, cb = split_whole_checkpoint(sd)
    assert bad_key[6:] not in nw and set(nw) == set(good) and list(tower) == ["shared.weight"] and cb.shape == (M, K, 8)
    assert split_whole_checkpoint(sd, not_load_document_encoder=True)[1] == {}


def test_real_sentencepiece_tokenizer_through_autotokenizer(tmp_path):
    """The CLIs tokenise with `AutoTokenizer.from_pretrained(dir)` of the installed `transformers`; the reference's vendored
    3.4 spelling `batch_encode_plus` no longer exists there.  `io.encode_batch` must produce the tokenizer contract of
    main_models.py:445-455 on a REAL SentencePiece model: ids, eos = 1, pad = 0 to max_length, mask over the real tokens."""
    pytest.importorskip("sentencepiece")
    from transformers import AutoTokenizer

    from mevi_amd.io import encode_batch
    from spm_fixture import build_t5_tokenizer_dir

    tok = AutoTokenizer.from_pretrained(build_t5_tokenizer_dir(str(tmp_path / "t5-ance")))
    texts = ["what is the capital of w3", "w10 w20", " ".join(f"w{i}" for i in range(60))]
    out = encode_batch(tok, texts, 32)
    ids, mask = out["input_ids"].numpy(), out["attention_mask"].numpy()
    assert ids.shape == mask.shape == (3, 32) and ids.dtype.kind == "i"
    for row, m in zip(ids, mask):
        n = int(m.sum())
        assert n >= 2 and m[:n].all() and not m[n:].any()            # right padding
        assert row[n - 1] == 1 and (row[n:] == 0).all() and (row[:n - 1] > 1).all()     # ... eos, then pad
    assert mask[2].all()                                            # truncated to the window, eos kept
    import sentencepiece as spm

    sp = spm.SentencePieceProcessor(model_file=str(tmp_path / "t5-ance" / "spiece.model"))
    assert ids[1, :int(mask[1].sum()) - 1].tolist() == sp.encode(texts[1])


def test_tower_directory_with_safetensors_only(tmp_path):
    """A newer export of the tower checkpoint (model.safetensors, no pytorch_model.bin) loads to the same tensors."""
    import json

    import torch
    from safetensors.torch import save_file

    from mevi_amd.evalrun import load_tower_weights

    sd = {"shared.weight": torch.randn(10, 8), "encoder.final_layer_norm.weight": torch.ones(8)}
    cfg = dict(d_model=8, d_ff=16, num_heads=2, d_kv=4, num_layers=1)
    for name in ("bin", "st"):
        os.makedirs(tmp_path / name)
        json.dump(cfg, open(tmp_path / name / "config.json", "w"))
    torch.save(sd, tmp_path / "bin" / "pytorch_model.bin")
    save_file(sd, str(tmp_path / "st" / "model.safetensors"))
    a, da = load_tower_weights(str(tmp_path / "bin"))
    b, db = load_tower_weights(str(tmp_path / "st"))
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a) and da.num_decoder_layers == db.num_decoder_layers == 1
    with pytest.raises(FileNotFoundError):
        os.makedirs(tmp_path / "none")
        json.dump(cfg, open(tmp_path / "none" / "config.json", "w"))
        load_tower_weights(str(tmp_path / "none"))


def test_file_position_of_memmap_views(tmp_path):
    """io.file_range: where a C-contiguous view of a mapped file starts IN THE FILE (upload_rows preads from there).  A sliced
    np.memmap reports its root's `.offset`: the position must follow the view."""
    from mevi_amd import io as mio

    a = np.arange(1000 * 8, dtype=np.float32).reshape(1000, 8)
    path = str(tmp_path / "x.bin")
    a.tofile(path)
    m = mio.map_rows(path, 8)
    assert m[10:20].offset == 0                                   # the numpy behaviour the function must not trust
    assert mio.file_range(m) == (path, 0)
    assert mio.file_range(m[17:517]) == (path, 17 * 32)
    assert mio.file_range(mio.map_rows(path, 8, first_row=3)[5:9]) == (path, 8 * 32)
    assert mio.file_range(np.memmap(path, dtype=np.float32, mode="r").reshape(-1, 8)[10:20]) == (path, 320)
    assert mio.file_range(a) is None and mio.file_range(m[:, :4]) is None and mio.file_range(m[::2]) is None
    for view in (m[17:517], mio.map_rows(path, 8, first_row=3)[5:9]):       # the bytes at that position are the view's bytes
        _, off = mio.file_range(view)
        with open(path, "rb") as f:
            f.seek(off)
            assert f.read(view.nbytes) == view.tobytes()


def test_context_image_exponent_and_bounds():
    """Host side of the split-image attention contexts (mevi_amd/ops.py): the exponent of a bound puts it in [2^14, 2^15) as
    pow2_exp of csrc/gemm_split.hip does for a row maximum, and the |V| bound is a true bound for rmsnorm / layernorm inputs."""
    import math

    from mevi_amd import ops

    for b in (1e-30, 3.7e-5, 0.999, 1.0, 1.0001, 37.0, 16384.0, 32768.0, 1e20):
        e = ops._pow2_exp(b)
        if -100 < e < 100:
            assert 2.0 ** 14 <= b * 2.0 ** e < 2.0 ** 15, (b, e)
    assert ops._pow2_exp(0.0) == 0 and ops._pow2_exp(float("inf")) == 0 and ops._pow2_exp(float("nan")) == 0
    assert ops._pow2_exp(1e-40) == 100 and ops._pow2_exp(1e38) == -100
    g = torch.Generator().manual_seed(3)
    d, inner = 96, 64
    ln_w, ln_b = torch.randn(d, generator=g), torch.randn(d, generator=g)
    wv, bv = torch.randn(inner, d, generator=g), torch.randn(inner, generator=g)
    x = torch.randn(50, d, generator=g) * torch.logspace(-3, 3, 50)[:, None]
    rms = x / torch.sqrt((x * x).mean(1, keepdim=True) + 1e-6) * ln_w                      # T5LayerNorm
    ln = torch.nn.functional.layer_norm(x, (d,), ln_w, ln_b, 1e-12)                          # BERT LayerNorm
    wmax = float(wv.norm(dim=1).max())
    assert float(rms.norm(dim=1).max()) <= ops.norm_out_bound(ln_w, d) * (1 + 1e-6)
    assert float(ln.norm(dim=1).max()) <= ops.norm_out_bound(ln_w, d, ln_b) * (1 + 1e-6)
    assert float((rms @ wv.T).abs().max()) <= ops.norm_out_bound(ln_w, d) * wmax * 1.001
    assert float((ln @ wv.T + bv).abs().max()) <= ops.norm_out_bound(ln_w, d, ln_b) * wmax * 1.001 + float(bv.abs().max())
    assert math.isclose(ops.norm_out_bound(ln_w, d), math.sqrt(d) * float(ln_w.abs().max()), rel_tol=1e-6)
    assert ops.ctx_bound(None, None) is None      # no bound, no image: the f32 form runs


def test_to_file_whole_matrix_path_writes_pythons_bytes(tmp_path):
    """faiss_search.to_file (MEVI/faiss_search.py:71-77) at a size that takes the one-call native renderer
    (mevi_format_ranked_rows: both list columns of every row, host threads sharing the rows): the same bytes as
    `','.join(str(x) for x in row.tolist())` per field -- incl. -1 padding ids, -FLT_MAX, signed zeros, subnormals, 1e16-style
    reprs -- and as the row-by-row path (fewer query lines than rows: the extra rows are not written, as the reference's zip)."""
    from mevi_amd import io as mio

    rng = np.random.default_rng(4)
    n, k = 90, 1000
    d = rng.standard_normal((n, k)).astype(np.float32)
    d[0, :8] = [0.0, -0.0, 1e-45, -3.4028235e38, 1e16, 123456.0, 1.5e-5, 16777216.0]
    i = rng.integers(-1, 8_841_823, (n, k))
    qp = tmp_path / "q.tsv"
    qp.write_text("".join(f"what is w{j} é\t{j}\n" for j in range(n - 3)))
    out = tmp_path / "dense.txt"
    mio.to_file(str(qp), str(out), d, i)
    want = "".join(f"what is w{j} é\t\t{','.join(str(x) for x in i[j].tolist())}\t{','.join(str(x) for x in d[j].tolist())}\n"
                   for j in range(n - 3))
    assert out.read_text() == want


def test_sentencepiece_t5_tokenizer_equals_the_hf_tokenizer(tmp_path):
    """io.load_tokenizer serves a T5 SentencePiece directory with SpmT5Tokenizer (no `transformers` import in generate.py /
    main.py): the ids and masks must be what `AutoTokenizer.from_pretrained(dir)(texts, max_length=L, padding='max_length',
    truncation=True, return_tensors='pt')` returns -- the call of MEVI/generate.py:85-87, MEVI/main_models.py:445-455 -- on plain,
    odd and over-long strings; a batch with a special-token spelling is handed to the HF object; MEVI_TOKENIZER=hf pins it."""
    pytest.importorskip("sentencepiece")
    from transformers import AutoTokenizer

    from mevi_amd import io as mio
    from spm_fixture import build_t5_tokenizer_dir

    d = build_t5_tokenizer_dir(str(tmp_path / "t5-ance"))
    hf = AutoTokenizer.from_pretrained(d)
    fast = mio.load_tokenizer(d)
    assert type(fast).__name__ == "SpmT5Tokenizer"
    rng = np.random.default_rng(0)
    texts = ["what is the capital of w5", "  leading and   multiple   spaces ", "", "w1", "UPPER case Words w7", "unicode é ü 中文 w3",
             "tab\there", "q1 " * 60, "x" * 500, "w1?w2!w3,", "new\nline", "<not a special>", "a < b > c"]
    texts += [" ".join(f"w{rng.integers(0, 300)}" for _ in range(rng.integers(1, 40))) for _ in range(200)]
    for L in (32, 8, 128):
        a = hf(texts, max_length=L, padding="max_length", truncation=True, return_tensors="pt")
        b = mio.encode_batch(fast, texts, L)
        assert torch.equal(a["input_ids"], b["input_ids"]) and torch.equal(a["attention_mask"], b["attention_mask"]), L
    special = ["has </s> inside", "plain", "an <extra_id_3> sentinel", "<pad> <unk>"]
    a = hf(special, max_length=16, padding="max_length", truncation=True, return_tensors="pt")
    b = mio.encode_batch(fast, special, 16)
    assert torch.equal(a["input_ids"], b["input_ids"]) and fast._hf is not None
    os.environ["MEVI_TOKENIZER"] = "hf"
    try:
        assert type(mio.load_tokenizer(d)).__name__ != "SpmT5Tokenizer"
    finally:
        del os.environ["MEVI_TOKENIZER"]


def test_cluster_index_sidecar_and_checkpoint_loader(tmp_path):
    """evalrun.load_cluster_index: the arrays beside `rqclus*.pkl` reproduce the index built from the pickle (document order inside
    a cluster included), are ignored when the pickle changed (size / mtime_ns fingerprint) or (M, K) differ, and are rewritten then;
    io.load_checkpoint reads zip-format, legacy-format and Lightning-style checkpoints (a Namespace beside the tensors)."""
    import argparse
    import pickle

    from mevi_amd import evalrun
    from mevi_amd import io as mio

    rng = np.random.default_rng(1)
    M, K = 3, 8
    cluster = {}
    for doc in rng.permutation(500).tolist():                       # lists in a non-sorted order: the sidecar must keep it
        cluster.setdefault(tuple(int(x) for x in rng.integers(0, K, M)), []).append(doc)
    p = str(tmp_path / "rqclus3_3.pkl")
    with open(p, "wb") as f:
        pickle.dump(cluster, f)
    a = evalrun.load_cluster_index(p, M, K)
    assert os.path.exists(evalrun.cluster_sidecar(p))
    b = evalrun.load_cluster_index(p, M, K)                         # from the arrays
    for x, y in ((a.keys, b.keys), (a.offsets, b.offsets), (a.doc_ids, b.doc_ids)):
        assert np.array_equal(x, y)
    key = next(iter(cluster))
    assert b.lookup(key).tolist() == cluster[key]
    cluster[key] = cluster[key][::-1] + [999]
    with open(p, "wb") as f:
        pickle.dump(cluster, f)
    os.utime(p, ns=(1, 1))                                          # even with an OLDER timestamp: a different fingerprint
    c = evalrun.load_cluster_index(p, M, K)
    assert c.lookup(key).tolist() == cluster[key]
    assert np.array_equal(evalrun.load_cluster_index(p, M, K).doc_ids, c.doc_ids)
    # checkpoints
    ck = str(tmp_path / "c.ckpt")
    w = {"a": torch.randn(7, 5)}
    torch.save({"state_dict": w, "hyper_parameters": argparse.Namespace(lr=1e-4)}, ck)
    sd = mio.load_checkpoint(ck)
    assert torch.equal(sd["state_dict"]["a"], w["a"]) and sd["hyper_parameters"].lr == 1e-4
    torch.save(w, ck, _use_new_zipfile_serialization=False)
    assert torch.equal(mio.load_checkpoint(ck)["a"], w["a"])


def test_phase_log_lines_and_fast_exit(tmp_path):
    """mevi_amd.phases: with MEVI_PHASE_LOG every mark appends `script<TAB>phase<TAB>seconds<TAB>since start`; finish() leaves the
    process at once with flushed streams (and is an ordinary return under MEVI_FAST_EXIT=0)."""
    import subprocess
    import sys

    log = str(tmp_path / "phases.log")
    code = ("import sys; sys.path.insert(0, %r); from mevi_amd import phases; phases.mark('imports'); print('out', flush=False); "
            "phases.mark('work'); phases.finish(); print('never')" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, MEVI_PHASE_LOG=log))
    assert r.returncode == 0 and r.stdout == "out\n", (r.stdout, r.stderr)
    lines = [l.split("\t") for l in open(log).read().splitlines()]
    assert [l[1] for l in lines] == ["imports", "work", "done"] and all(float(l[2]) >= 0 and float(l[3]) > 0 for l in lines)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, MEVI_FAST_EXIT="0"))
    assert r.returncode == 0 and r.stdout == "out\nnever\n"
