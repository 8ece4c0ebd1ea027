"""Eval driver of the seq2seq arm: what `python main.py --mode eval ...` (marco_eval_nci_rq.sh)
does in the reference -- main.inference / partial_inference (MEVI/main.py:267-337),
T5FineTunerWithValidation.on_validation_epoch_start / infer / validation_epoch_end
(MEVI/main_models.py:4215-4273, 3555-4098, 4290-4393) and handle_infer_results (:4100-4201).

Per process (one per GPU, queries split by rank like DistributedSampler(shuffle=False)):
  NCI beam search (mevi_amd.nci) -> coarse log + ranks -> twin-tower query embedding (mevi_amd.t5)
  -> in-cluster re-ranking (mevi_amd.fine) -> fine / hard-negative logs -> rank-merged TSVs + metrics.
"""
import os
import pickle
import time

import numpy as np
import pandas as pd
import torch

from . import dense as mdense
from . import fine as mfine
from . import hip
from .io import RankLog, encode_batch, join_i64, load_checkpoint, upload_rows
from .phases import mark
from .nci import MODEL_INFO, NCIModel, check_weights, config_from_weights, decode_token
from .rq import ClusterIndex, ProductQuantization
from .t5 import T5Dims, TwinTower


# ---- loading -------------------------------------------------------------------------------
MODEL_PREFIXES = ("shared.", "encoder.", "decoder.", "lm_head.", "decode_embeddings.", "adaptor")
BAD_WHOLE_KEYS = (   # dropped by try_load_ckpt's whole-model branch (MEVI/main.py:207-214)
    "model.decoder.block.0.layer.1.EncDecAttention.relative_attention_bias.weight",
    "model.ori_decoder.block.0.layer.1.EncDecAttention.relative_attention_bias.weight",
    "document_encoder.lm_q.decoder.block.0.layer.1.EncDecAttention.relative_attention_bias.weight",
    "document_encoder.lm_p.decoder.block.0.layer.1.EncDecAttention.relative_attention_bias.weight")


def _state_dict(path):
    sd = load_checkpoint(path)
    return sd["state_dict"] if "state_dict" in sd else sd


def nci_weights_from_state_dict(sd, report=print):
    """try_load_ckpt's nci_path branch (MEVI/main.py:232-248): strip the Lightning 'model.' prefix; a key the NCI model
    does not have (another module's tensors inside the checkpoint) is reported as `Bad parameter <k>.` and skipped, as
    the reference does.  Shape mismatches are judged by nci.check_weights once the model's dimensions are known."""
    out = {}
    for k, v in sd.items():
        if k.startswith("model."):
            k = k[6:]
        if not torch.is_tensor(v):
            continue
        if not k.startswith(MODEL_PREFIXES):
            report(f"Bad parameter {k}.")
            continue
        out[k] = v
    if "lm_head.weight" not in out and "decode_embeddings.weight" in out:  # tie_decode_embedding=1
        out["lm_head.weight"] = out["decode_embeddings.weight"]
    return out


def load_nci_weights(path, report=print):
    """The NCI checkpoint: a Lightning ckpt ('state_dict' with 'model.' prefixes) or a bare state dict."""
    return nci_weights_from_state_dict(_state_dict(path), report)


def split_whole_checkpoint(sd, not_load_document_encoder=False):
    """try_load_ckpt's whole-model branch (--infer_ckpt, MEVI/main.py:203-230): the state dict of the full
    T5FineTunerWithValidation -> (NCI weights, query-tower overrides, RQ codebook or None).  The four
    relative_attention_bias keys the reference filters are dropped; with not_load_document_encoder the
    `document_encoder.` tensors are left at what the tower directory holds."""
    sd = {k: v for k, v in sd.items() if k not in BAD_WHOLE_KEYS and torch.is_tensor(v)}
    nci_w = {k[6:]: v for k, v in sd.items() if k.startswith("model.")}
    if "lm_head.weight" not in nci_w and "decode_embeddings.weight" in nci_w:
        nci_w["lm_head.weight"] = nci_w["decode_embeddings.weight"]
    pre = "document_encoder.lm_q."
    tower = {} if not_load_document_encoder else {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    return nci_w, tower, sd.get("pq.codebook")


def load_hf_state_dict(model_dir):
    """The weights of an HF model directory: `pytorch_model.bin` (what the reference's `from_pretrained` of 2023 reads), or
    `model.safetensors` when a newer export of the same checkpoint ships only that."""
    path = os.path.join(model_dir, "pytorch_model.bin")
    if os.path.exists(path):
        return load_checkpoint(path)
    st = os.path.join(model_dir, "model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file

        return load_file(st, device="cpu")
    raise FileNotFoundError(f"{model_dir}: neither pytorch_model.bin nor model.safetensors")


def load_tower_weights(model_dir):
    """T5-ANCE directory in HF layout (config.json + pytorch_model.bin), as DocumentEncoder.build ->
    AutoModel.from_pretrained reads it (MEVI/document_encoder.py:176-188)."""
    import json

    cfg = json.load(open(os.path.join(model_dir, "config.json")))
    sd = load_hf_state_dict(model_dir)
    if "encoder.embed_tokens.weight" in sd and "shared.weight" not in sd:
        sd["shared.weight"] = sd["encoder.embed_tokens.weight"]
    dims = T5Dims(d_model=cfg["d_model"], d_ff=cfg["d_ff"], num_heads=cfg["num_heads"], d_kv=cfg["d_kv"],
                  num_layers=cfg["num_layers"], num_decoder_layers=cfg.get("num_decoder_layers", cfg["num_layers"]),
                  layer_norm_epsilon=cfg.get("layer_norm_epsilon", 1e-6),
                  relative_attention_num_buckets=cfg.get("relative_attention_num_buckets", 32))
    return sd, dims


def load_bert_tower(model_path, device, batch_size=None):
    """BERT-family tower as `generate.get_document_encoder` + `DocumentEncoder.build` load it (MEVI/generate.py:31-44,
    document_encoder.py:141-188): an HF directory (config.json + pytorch_model.bin, tied towers), or an AR2 checkpoint
    `ar2g_{nq,marco}_finetune.pkl` ({'model_dict': {'ctx_model.*', 'question_model.*'}}) with its config directory
    next to it (`ernie-2.0-base-en` / `co-condenser-marco-retriever`)."""
    import json

    from .bert import BertTower

    def strip(sd):  # AutoModel state dicts may carry the task-model prefix
        for pre in ("bert.", "ernie."):
            if any(k.startswith(pre + "embeddings.") for k in sd):
                return {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
        return sd

    wp = None
    if model_path.endswith(".pkl"):
        cfg_dir = os.path.join(os.path.split(model_path)[0],
                               "ernie-2.0-base-en" if model_path.endswith("ar2g_nq_finetune.pkl")
                               else "co-condenser-marco-retriever")
        params = load_checkpoint(model_path)["model_dict"]
        wq = strip({k[len("question_model."):]: v for k, v in params.items() if k.startswith("question_model.")})
        wp = strip({k[len("ctx_model."):]: v for k, v in params.items() if k.startswith("ctx_model.")})
    else:
        cfg_dir = model_path
        wq = strip(load_hf_state_dict(model_path))
    cfg = json.load(open(os.path.join(cfg_dir, "config.json")))
    return BertTower(wq, cfg["num_hidden_layers"], cfg["num_attention_heads"], weights_p=wp,
                     eps=cfg.get("layer_norm_eps", 1e-12), device=device, batch_size=batch_size)


def tower_model_type(model_path):
    """'t5' or 'bert' (BERT / ERNIE), from the config.json the reference's AutoConfig would read."""
    import json

    if model_path.endswith(".pkl"):
        return "bert"
    mt = json.load(open(os.path.join(model_path, "config.json"))).get("model_type", "t5")
    return "bert" if mt in ("bert", "ernie") else "t5"


def load_queries(data_dir, n_test=-1, fname="dev_mevi_dedup.tsv"):
    """dev_mevi_dedup.tsv: `query \\t id,id,...` (main_utils.load_data_infer, MEVI/main_utils.py:271-278)."""
    df = pd.read_csv(os.path.join(data_dir, fname), names=["query", "oldid"], encoding="utf-8", header=None,
                     sep="\t", converters={"oldid": lambda x: [int(v) for v in x.split(",")]})
    assert not df.isnull().values.any()
    if n_test is not None and n_test >= 0:
        df = df[:n_test]
    return df


def load_nq_queries(data_dir, n_test=-1, fname="nq-test.qa.csv"):
    """--dataset nq_dpr: `question \t answers` (main_utils.py:279-287).  The 'oldid' column carries the row index -- the
    `query_indices` a sample is judged by (main_models.py:857-858)."""
    df = pd.read_csv(os.path.join(data_dir, fname), names=["query", "answers"], encoding="utf-8", header=None, sep="\t")
    assert not df.isnull().values.any()
    df["oldid"] = np.arange(len(df))
    if n_test is not None and n_test >= 0:
        df = df[:n_test]
    return df


class CodeMap:
    """`pq_mapping[doc] -> code tuple` (main_models.py:815-831) over an i32 [N, M] array."""

    def __init__(self, codes):
        self.codes = codes

    def __getitem__(self, doc):
        c = self.codes[doc]
        if c[0] < 0:
            raise KeyError(doc)
        return tuple(c.tolist())

    def __len__(self):
        return len(self.codes)


class NqAnswers:
    """test_inverse_offsets.bin / test_inverse_array.bin (i32): per document, the test questions it answers
    (main_models.py:4267-4272).  A hit test `qind in array[offsets[d]:offsets[d+1]]` per ranked document is what the
    reference loops over (:3744-3750, :4064-4069); here the lists are inverted once (question -> sorted doc ids) and a
    ranked list is tested with one `np.isin`."""

    def __init__(self, data_dir):
        offsets = np.asarray(np.memmap(os.path.join(data_dir, "test_inverse_offsets.bin"), mode="r", dtype=np.int32), np.int64)
        array = np.asarray(np.memmap(os.path.join(data_dir, "test_inverse_array.bin"), mode="r", dtype=np.int32))
        doc = np.repeat(np.arange(len(offsets) - 1, dtype=np.int64), np.diff(offsets))
        order = np.argsort(array[:offsets[-1]], kind="stable")
        self.question, self.doc = array[:offsets[-1]][order], doc[order]

    def docs_answering(self, qind):
        lo, hi = np.searchsorted(self.question, [qind, qind + 1])
        return self.doc[lo:hi]

    def first_hit(self, qind, ranked_docs):
        hit = np.flatnonzero(np.isin(np.asarray(ranked_docs, dtype=np.int64), self.docs_answering(qind)))
        return int(hit[0]) if len(hit) else None


def _corpus_width_hint(a):
    """Row width of `--embedding_path` when it can be known before the tower is loaded: the T5-ANCE directory's config.json
    (`d_model`); None for the BERT-family towers and for anything unreadable."""
    import json

    mode = os.environ.get("MEVI_OVERLAP_UPLOAD", "1")          # "0": never, "always": also for small files (tests)
    if (getattr(a, "document_encoder", None) or "ance") != "ance" or mode == "0":
        return None
    try:
        with open(os.path.join(a.ckpt_dir, "t5-ance", "config.json")) as f:
            d = int(json.load(f)["d_model"])
        size = os.path.getsize(a.embedding_path)
    except (OSError, ValueError, KeyError, TypeError):
        return None
    return d if d > 0 and (size >= (256 << 20) or mode == "always") and size % (4 * d) == 0 else None      # small files: nothing to overlap


def cluster_sidecar(pq_cluster_path):
    return pq_cluster_path + ".index.npz"


def write_cluster_sidecar(pq_cluster_path, index):
    """The cluster dict of `rqclus*.pkl` as three arrays beside it (keys, offsets, document ids IN THE PICKLE'S ORDER) + the
    pickle's fingerprint: unpickling the 1 M-key / 8.8 M-id dict and flattening it took main.py 1.35 s of every start
    (tools/e2e_cli.py); the arrays load in ~50 ms.  Best effort (a read-only directory keeps the pickle path)."""
    from .metrics import _pickle_fingerprint

    try:
        fp = _pickle_fingerprint(pq_cluster_path)
        tmp = cluster_sidecar(pq_cluster_path) + ".tmp.npz"
        np.savez(tmp, keys=index.keys, offsets=index.offsets, doc_ids=index.doc_ids,
                 meta=np.array([index.M, index.K, fp["size"], fp["mtime_ns"]], np.int64))
        os.replace(tmp, cluster_sidecar(pq_cluster_path))
    except OSError:
        pass


def load_cluster_index(pq_cluster_path, M, K, write=True):
    """ClusterIndex of `rqclus*.pkl` (MEVI/main_models.py:3200-3203): from the array sidecar when it was made for THIS pickle
    (size and mtime_ns) and this (M, K), else from the pickle -- and the sidecar is (re)written for the next start."""
    from .metrics import _pickle_fingerprint

    side = cluster_sidecar(pq_cluster_path)
    if os.path.exists(side) and os.environ.get("MEVI_CLUSTER_SIDECAR", "1") != "0":
        try:
            fp = _pickle_fingerprint(pq_cluster_path)
            with np.load(side) as z:
                if z["meta"].tolist() == [M, K, fp["size"], fp["mtime_ns"]]:
                    return ClusterIndex(M, K, z["keys"], z["offsets"], z["doc_ids"])
        except (OSError, ValueError, KeyError):
            pass
    with open(pq_cluster_path, "rb") as f:
        index = ClusterIndex.from_dict(pickle.load(f), M, K)
    if write:
        write_cluster_sidecar(pq_cluster_path, index)
    return index


def rank_slice(n, rank, nrank):
    """Indices DistributedSampler(shuffle=False) gives `rank`: every nrank-th item of the list padded to
    a multiple of nrank by repeating its head (MEVI/main.py:318-322); duplicates are de-duplicated by the
    query-keyed dicts downstream."""
    total = (n + nrank - 1) // nrank * nrank
    idx = list(range(n)) + list(range(total - n))
    return idx[rank:total:nrank]


class EvalRun:
    def __init__(self, args, tokenizer=None, rank=0, nrank=1, barrier=None, device=None, tower_tokenizer=None):
        self.args, self.rank, self.nrank = args, rank, nrank
        self.barrier = barrier or (lambda: None)
        hip.require_gpu()
        self.dev = torch.device(device if device is not None else "cuda")
        a = args
        self.M, self.K, self.R = a.subvector_num, 2 ** a.subvector_bits, a.num_return_sequences
        tower_override, ckpt_codebook = {}, None
        # The 27 GB corpus upload (file reads + PCIe: ~0.8 s, 8 reader threads) and the checkpoint loads / model build below
        # (one Python thread: ~0.8 s) need different resources: the upload starts NOW on a background thread and is joined
        # where `self.emb` is first needed.  Only when the row width is known without loading the tower (T5-ANCE: config.json).
        emb_job = self._start_corpus_upload(a)
        if getattr(a, "infer_ckpt", None):        # whole-model checkpoint (MEVI/main.py:203-230) takes precedence
            nci_w, tower_override, ckpt_codebook = split_whole_checkpoint(
                _state_dict(a.infer_ckpt), bool(getattr(a, "not_load_document_encoder", 0)))
        else:
            nci_w = load_nci_weights(a.nci_ckpt)
        self.cfg = config_from_weights(nci_w, self.M, self.K)
        info = MODEL_INFO.get(getattr(a, "model_info", None))
        if info is not None and info[3] == self.cfg.d_model:
            # --model_info defines the model the reference builds (MEVI/main.py:755-773); a checkpoint of another width
            # (the miniature models of the tests) defines its own
            self.cfg.num_layers, self.cfg.num_decoder_layers, self.cfg.d_ff = info[0], info[1], info[2]
            self.cfg.adaptor_layers = getattr(a, "adaptor_layer_num", self.cfg.adaptor_layers)
        bad = check_weights(nci_w, self.cfg)
        if bad:
            raise SystemExit(f"{len(bad)} tensor(s) of the NCI model are missing from the checkpoint or have another shape "
                             f"(first: {bad[0]}): the reference would leave them at their random initialisation and go on "
                             f"(MEVI/main.py:243-246); check --subvector_num / --subvector_bits / --model_info / "
                             f"--adaptor_layer_num against the checkpoint")
        d_model = self.cfg.d_model
        self.nci = NCIModel(nci_w, cfg=self.cfg, device=self.dev)
        del nci_w
        mark("NCI checkpoint -> model in HBM", sync=True)
        # the tower and its tokenizer (init_document_encoder, MEVI/main_models.py:1643-1681)
        enc = getattr(a, "document_encoder", None) or "ance"
        tower_dir = os.path.join(a.ckpt_dir, "t5-ance")       # NCI shares the T5-ANCE vocabulary in every configuration
        if enc == "ance":
            tw, tdims = load_tower_weights(tower_dir)
            tw.update(tower_override)             # document_encoder.lm_q.* of a whole-model checkpoint
            self.tower = TwinTower(tw, dims=tdims, device=self.dev)
        elif enc == "cocondenser":
            self.tower = load_bert_tower(os.path.join(a.ckpt_dir, "co-condenser-marco-retriever"), self.dev)
        elif enc == "ar2":
            name = "ar2g_marco_finetune.pkl" if getattr(a, "dataset", "marco") == "marco" else "ar2g_nq_finetune.pkl"
            self.tower = load_bert_tower(os.path.join(a.ckpt_dir, name), self.dev)
        else:
            raise NotImplementedError(enc)
        if tokenizer is None:
            from .io import load_tokenizer      # host-side tokenisation: the HF tokenizer, or SentencePiece itself for T5 directories

            tokenizer = load_tokenizer(tower_dir)
        self.tokenizer = tokenizer
        mark("tower weights + tokenizer", sync=True)
        # BERT-family towers read the query through their own tokenizer (`qenc_source_ids`, main_models.py:853-856):
        # bert-base-uncased, special tokens only for 'ar2' (main_models.py:359-360)
        self.tower_tokenizer = tower_tokenizer
        self.tower_special_tokens = enc in ("ance", "ar2")
        if enc != "ance" and tower_tokenizer is None:
            from transformers import AutoTokenizer

            local = os.path.join(a.ckpt_dir, "bert-base-uncased")
            self.tower_tokenizer = AutoTokenizer.from_pretrained(local if os.path.isdir(local) else "bert-base-uncased",
                                                                 do_lower_case=True)
        # corpus embeddings resident in HBM (the reference keeps a CPU memmap and copies per cluster)
        d_model = self.tower.dim     # the corpus embeddings and the RQ codebook live in the tower's output space
        n_docs = os.path.getsize(a.embedding_path) // (4 * d_model)
        if emb_job is not None and emb_job[1] == d_model:
            emb_job[0].join()
            if "error" in emb_job[2]:
                raise emb_job[2]["error"]
            self.emb = emb_job[2]["emb"]
        else:
            if emb_job is not None:               # the tower's width is not what its config.json said: upload again, the right shape
                emb_job[0].join()
                emb_job[2].clear()
            emb = np.memmap(a.embedding_path, dtype=np.float32, mode="r", shape=(n_docs, d_model))
            self.emb = upload_rows(emb, self.dev)
        mark("corpus embeddings file -> HBM (rest after the overlap with the model loads)" if emb_job is not None else "corpus embeddings file -> HBM")
        # RQ codebook + cluster index (pickles if present, else encode on the GPU and write them)
        self.pq = ProductQuantization("rq", self.M, a.subvector_bits, "l2", d_model, device=self.dev)
        if ckpt_codebook is not None:             # --infer_ckpt carries pq.codebook: pq.initialize is skipped (main_models.py:4252)
            self.pq.load_codebook(ckpt_codebook)
        else:
            self.pq.initialize(a.pq_path, rank=0)
        map_path = a.pq_cluster_path.replace("clus", "mapping")
        # every rank looks BEFORE anyone writes (barrier), so all ranks take the same branch and the barriers below pair
        # up: a rank arriving after rank 0 had written the pickles used to skip rank 0's barrier (one-off from then on)
        have_clusters = os.path.exists(a.pq_cluster_path) and os.path.exists(map_path)
        self.barrier()
        if have_clusters:
            self.index = load_cluster_index(a.pq_cluster_path, self.M, self.K, write=(rank == 0))
            # rqmapping*.pkl is the inverse of the cluster dict (gen_pq_doc_cluster writes both from one encode); it is
            # rebuilt from the index as an array instead of unpickling 8.8 M tuples (9 s + 2 GB per rank on MS MARCO)
        else:
            self.index = self.pq.get_document_cluster(self.emb, 0, 1, as_index=True)
            if rank == 0:
                cluster, mapping = self.index.to_dicts()
                with open(a.pq_cluster_path, "wb") as f:
                    pickle.dump(cluster, f)
                with open(map_path, "wb") as f:
                    pickle.dump(mapping, f)
                del cluster, mapping
                write_cluster_sidecar(a.pq_cluster_path, self.index)
                from .metrics import write_mapping_sidecar

                write_mapping_sidecar(map_path, self.index.doc_codes(n_docs))   # the same mapping as an array (ensemble scripts)
        self.barrier()
        self.mapping = CodeMap(self.index.doc_codes(n_docs))
        mark("RQ codebook + cluster index (%s)" % ("pickle read" if have_clusters else "encode on the GPU + pickles written"), sync=True)
        print("Number of all pq document clusters:", len(self.index.keys))
        # --doc_multiclus C > 1 (gen_pq_doc_topk, main_models.py:3222-3262): every document also belongs to the clusters of
        # its top-C code paths; rqtopk<C>*.pt holds the paths, rqmulticlus<C>*.pkl the code -> documents dict
        self.C = int(getattr(a, "doc_multiclus", 1) or 1)
        self.aggregate = None
        if self.C > 1:
            self.aggregate = getattr(a, "multiclus_score_aggr", "add")
            topk_path = a.pq_cluster_path.replace("clus", f"topk{self.C}").replace(".pkl", ".pt")
            multi_path = topk_path.replace("topk", "multiclus").replace(".pt", ".pkl")
            have_topk = os.path.exists(topk_path)
            self.barrier()                                    # same rule: look first, then write, then meet again
            if not have_topk:
                labels = self.pq.get_topk_document_mapping(self.emb, 0, 1, self.C)
                if rank == 0:
                    torch.save(labels, topk_path)
            self.barrier()
            self.doc_topk = torch.load(topk_path, map_location="cpu").numpy()
            self.index = ClusterIndex.from_topk_labels(self.doc_topk, self.K)
            if rank == 0 and not os.path.exists(multi_path):
                with open(multi_path, "wb") as f:
                    pickle.dump(self.index.to_dicts()[0], f)
            self.barrier()
        self.fine = mfine.FineStage(self.emb, self.index)
        self._dense_index = None
        # --query_embedding_path (this build): the file `generate.py --gen_query` wrote for the same query file.  The
        # reference encodes every query a second time inside infer() (main_models.py:3797-3812); the embeddings are the
        # same bits either way (deterministic, grouping-independent kernels), so the fine stage can read them.
        self.query_table = None
        qpath = getattr(a, "query_embedding_path", None)
        if qpath:
            self.query_table = np.memmap(qpath, dtype=np.float32, mode="r").reshape(-1, d_model)
        # --dataset nq_dpr: samples carry their question index instead of gt doc ids; hits are answer-based
        self.nq = NqAnswers(a.data_dir) if getattr(a, "dataset", "marco") == "nq_dpr" else None
        prefix = a.custom_save_path[:-4]
        # --eval_all_documents 1 (recall_level 'fine'): the brute-force ablation -- no beam search, the fine list is the
        # exact top-max(recall_num) of q.d over the whole corpus (main_models.py:3570,3818-3876)
        self.eval_all = bool(getattr(a, "eval_all_documents", 0))
        # --timing_infer_step N: per-step wall clock of the beam search ('nci') and of the fine stage ('knn'), pickled to
        # times<R>.pkl after N + 1 steps (main_models.py:1250-1252,4091-4096); the towers and the search then replay HIP graphs
        self.timer = {"nci": [], "knn": []} if getattr(a, "timing_infer_step", 0) > 0 else None
        self.timing_step_for_infer = 0
        # --use_topic_model 1 (topic_score_ratio 0, doc_multiclus 1): document score = NCI score of its cluster x q.d
        # (get_inference_scores, main_models.py:3539-3552) -- the beam scores on the cluster path, the scores of ALL
        # code paths (_generate_all) with --eval_all_documents
        self.topic = bool(getattr(a, "use_topic_model", 0))
        if self.topic:
            assert self.C == 1 or not self.eval_all, "use_topic_model over multi-cluster documents is built for the cluster path"
            self.ratio = float(getattr(a, "topic_score_ratio", 0) or 0)
            self.doc_path = None
            # topic_score_ratio > 0 (`additional_reconstruct`, main_models.py:1270): doc_proba[d] = <reconstruct(codes(d)), emb[d]>
            # (gen_all_reconstruct :3272-3307 + gen_doc2index_mapping :3360-3364, bmm form of compute_similarity); with
            # multi-cluster documents the probability is per (document, cluster) and computed per candidate (FineStage.rerank)
            self.doc_proba = self._doc_proba() if self.ratio and self.C == 1 else None
        # --recall_level: 'both' (the eval scripts), 'coarse' (beam clusters only: no tower pass, no fine stage) or 'fine'
        # (fine list only; main_models.py:3736,3781,4103-4110)
        self.level = "fine" if self.eval_all else getattr(a, "recall_level", "both")
        self.coarse_log = RankLog(f"{prefix}_coarse.tsv", rank, nrank, self.barrier) if self.level in ("coarse", "both") else None
        self.fine_log = RankLog(f"{prefix}_fine.tsv", rank, nrank, self.barrier) if self.level in ("fine", "both") else None
        self.hn_log = RankLog(f"{prefix}_hn{a.save_hard_neg}.tsv", rank, nrank, self.barrier) if a.save_hard_neg else None

    def tokenize(self, queries):
        out = encode_batch(self.tokenizer, queries, 32)
        return out["input_ids"], out["attention_mask"]

    def query_embedding(self, texts, ids, mask, rows=None):
        if self.query_table is not None:     # --query_embedding_path: generate.py already encoded these queries
            return torch.from_numpy(np.ascontiguousarray(self.query_table[np.asarray(rows)])).to(self.dev)
        kw = {"graph": True} if self.timer is not None and isinstance(self.tower, TwinTower) else {}
        if self.tower_tokenizer is None:      # T5-ANCE: the tower reads the NCI input ids (main_models.py:3797-3799)
            return self.tower.encode_query({"input_ids": ids, "attention_mask": mask}, **kw)
        tok = encode_batch(self.tower_tokenizer, texts, 32, add_special_tokens=self.tower_special_tokens)
        return self.tower.encode_query({"input_ids": tok["input_ids"], "attention_mask": tok["attention_mask"]}, **kw)

    @torch.no_grad()
    def infer_all_documents(self, texts, doc_ids, ids, mask, rows=None):
        """--eval_all_documents: [(text, N, fine ranks)].  The reference streams the corpus in --encode_batch_size
        blocks through a running top-pool (main_models.py:3818-3876); the result is the exact top-pool of q.d, which
        is what the dense arm's search returns (ties by ascending id here, unspecified `torch.topk` order there).
        The hard-negative line keeps the reference's quirk (:3905-3908): its score column is NOT the sorted scores but
        the last iteration's concatenation -- the running top-pool before the last block, then the last block's raw
        scores in id order -- cut to save_hard_neg."""
        a = self.args
        N = self.emb.shape[0]
        pool = max(a.recall_num)
        qemb = self.query_embedding(texts, ids, mask, rows)
        if self.topic:
            return self._all_documents_topic(texts, doc_ids, ids, mask, qemb, pool)
        if self._dense_index is None:
            self._dense_index = mdense.DenseIndex(self.emb)
        _, top_i = self._dense_index.search(qemb, min(N, pool))
        top_i = top_i.cpu().numpy()
        if self.hn_log is not None:
            bs = max(1, a.encode_batch_size or 64)
            last = ((N - 1) // bs) * bs
            tail_s, tail_i = mdense.ip_topk(qemb, self.emb[last:], N - last, id_offset=last)
            tail = torch.gather(tail_s, 1, torch.argsort(tail_i, dim=1))         # the last block's scores in id order
            if last > 0:
                head_s, _ = mdense.ip_topk(qemb, self.emb[:last], min(last, pool))
                tail = torch.cat([head_s, tail], dim=1)
            quirk = tail.cpu().numpy()
            gt_s = self.fine.gt_scores(qemb, doc_ids) if self.nq is None else None
        results = []
        for i, text in enumerate(texts):
            docs = top_i[i]
            self.fine_log.add((text, docs.tolist(), doc_ids[i]) if self.nq is None else (text, docs.tolist()))
            if self.hn_log is not None:
                n = a.save_hard_neg
                self.hn_log.add((text, mfine.f32_repr(gt_s[i]) if self.nq is None else "", join_i64(docs[:n]),
                                 mfine.f32_repr(quirk[i][:n])))
            ranks = mfine.fine_ranks(docs, doc_ids[i]) if self.nq is None else [self.nq.first_hit(doc_ids[i], docs)]
            results.append((text, N, ranks))
        return results

    def _doc_proba(self, chunk=1 << 20):
        """<sum_j codebook[j][code_j(d)] (level 0 first), emb[d]> for every document, f32 [N] (exact fmaf chains)."""
        from . import ops

        N = self.emb.shape[0]
        codes = torch.from_numpy(self.index.doc_codes(N).astype(np.int64)).to(self.dev)
        cb = self.pq.get_codebook().to(self.dev)
        out = torch.empty(N, dtype=torch.float32, device=self.dev)
        for a in range(0, N, chunk):
            c = codes[a:a + chunk]
            rec = cb[0][c[:, 0]]
            for j in range(1, self.M):
                rec = rec + cb[j][c[:, j]]
            idx = torch.arange(c.shape[0], dtype=torch.int64, device=self.dev)
            out[a:a + chunk] = ops.pair_dot(rec.contiguous(), idx, self.emb[a:a + chunk], idx)
        return out

    def _all_documents_topic(self, texts, doc_ids, ids, mask, qemb, pool):
        """--use_topic_model 1 --eval_all_documents 1 (main_models.py:3565,3653-3656,3818-3876): score(q, d) =
        all_scores[q, path(d)] * (q.d), all_scores = the NCI scores of all K**M code paths (_generate_all), path(d) the
        document's RQ code as a mixed-radix index (gen_doc2index_mapping :3311-3372, kary = K); streamed over the corpus
        in --encode_batch_size blocks through a running top-pool.  Order: score desc, ties by ascending id (torch.topk
        leaves ties unspecified)."""
        from . import ops

        a, N, K, M = self.args, self.emb.shape[0], self.K, self.M
        _, all_scores, _, _ = self.nci.generate(ids, mask, num_beams=1, num_return_sequences=1,
                                                length_penalty=a.length_penalty, eval_all_documents=True)
        if self.doc_path is None:
            codes = torch.from_numpy(self.index.doc_codes(N).astype(np.int64)).to(self.dev)
            self.doc_path = sum(codes[:, p] * K ** (M - 1 - p) for p in range(M))
        B = qemb.shape[0]
        bs = max(1, min(a.encode_batch_size or 64, 8192))
        kk = min(pool, N)
        run_s = torch.full((B, kk), -torch.finfo(torch.float32).max, dtype=torch.float32, device=self.dev)
        run_i = torch.full((B, kk), -1, dtype=torch.int64, device=self.dev)
        last_scores = None
        for start in range(0, N, bs):
            end = min(N, start + bs)
            qd = ops.linear(qemb, self.emb[start:end])
            if self.ratio:
                qd = self.ratio * self.doc_proba[None, start:end] + (1 - self.ratio) * qd
            new = all_scores[:, self.doc_path[start:end]] * qd                                        # f32 tensor ops, this order
            if self.hn_log is not None and end == N:
                filled = min(start, kk)                                         # rows seen so far, capped by the pool
                last_scores = torch.cat([run_s[:, :filled], new], dim=1)        # the reference's `scores` (:3872,3905)
            width = max(kk, end - start)
            pad = lambda t, v: torch.nn.functional.pad(t, (0, width - t.shape[1]), value=v)   # noqa: E731
            ids_new = torch.arange(start, end, device=self.dev)[None].expand(B, -1)
            ls = torch.stack([pad(run_s, -torch.finfo(torch.float32).max), pad(new, -torch.finfo(torch.float32).max)])
            li = torch.stack([pad(run_i, -1), pad(ids_new, -1)])
            run_s, run_i = mdense.topk_merge(ls, li, kk)
        top_i = run_i.cpu().numpy()
        gt_s = quirk = None
        if self.hn_log is not None:
            quirk = last_scores.cpu().numpy()
            gt_s = self.fine.gt_scores(qemb, doc_ids) if self.nq is None else None
        results = []
        for i, text in enumerate(texts):
            docs = top_i[i]
            self.fine_log.add((text, docs.tolist(), doc_ids[i]) if self.nq is None else (text, docs.tolist()))
            if self.hn_log is not None:
                n = a.save_hard_neg
                self.hn_log.add((text, mfine.f32_repr(gt_s[i]) if self.nq is None else "", join_i64(docs[:n]),
                                 mfine.f32_repr(quirk[i][:n])))
            ranks = mfine.fine_ranks(docs, doc_ids[i]) if self.nq is None else [self.nq.first_hit(doc_ids[i], docs)]
            results.append((text, N, ranks))
        return results

    def decode_tree(self):
        """The tree the beams are held to.  `--codebook 1` (every script) = the shared-sons tree of all K codes per level
        (main_models.py:1698-1706): None.  MEVI_DECODE_TREE=clusters (a switch of this build; the reference reaches the same
        structure only outside --codebook mode, main_models.py:1707-1728): the generic trie of the code paths that own a
        populated cluster, TreeBuilder(share_sons=False) -- no beam is spent on an empty cluster."""
        if os.environ.get("MEVI_DECODE_TREE", "") != "clusters":
            return None
        if getattr(self, "_decode_tree", None) is None:
            from .nci import PrefixTree

            keys = np.asarray(self.index.keys, dtype=np.int64)
            w = self.K ** np.arange(self.M - 1, -1, -1, dtype=np.int64)
            self._decode_tree = PrefixTree((keys[:, None] // w) % self.K, self.M, self.K, self.dev)
        return self._decode_tree

    @torch.no_grad()
    def infer(self, texts, doc_ids, rows=None):
        """One batch: returns [(text, ndoc, coarse ranks, fine ranks)] like infer() with recall_level='both'.
        rows: the samples' line numbers in the query file (only read with --query_embedding_path)."""
        a, R = self.args, self.R
        ids, mask = self.tokenize(texts)
        if self.eval_all:
            return self.infer_all_documents(texts, doc_ids, ids, mask, rows)
        timing = self.timer is not None        # --timing_infer_step (main_models.py:3558,3729-3732,4057-4059,4091-4096)
        if timing:
            torch.cuda.synchronize()
            t0 = time.time()
        decoded, scores, _, _ = self.nci.generate(ids, mask, num_beams=R, num_return_sequences=R,
                                                  length_penalty=a.length_penalty, max_length=self.M + 2, graph=timing,
                                                  decode_tree=self.decode_tree())
        B = len(texts)
        codes = decode_token(decoded, self.K).view(B, R, self.M).cpu().numpy()
        scores = np.array(scores).reshape(B, R)
        if timing:
            t1 = time.time()
            self.timer["nci"].append(t1 - t0)
        want_c, want_f = self.level in ("coarse", "both"), self.level in ("fine", "both")
        if not want_f:      # recall_level 'coarse': cluster ranks and the candidate count only (main_models.py:3736-3780)
            ndoc = self.fine.candidates_device(codes)[3]
            results = []
            for i, text in enumerate(texts):
                d = codes[i].tolist()
                cr, gt_codes = self._coarse_ranks(d, doc_ids[i])
                self.coarse_log.add((text, d, gt_codes, scores[i].tolist()) if self.nq is None else (text, d, scores[i].tolist()))
                results.append((text, int(ndoc[i]), cr))
            return self._timed(results, t1 if timing else None)
        qemb = self.query_embedding(texts, ids, mask, rows)
        weights = None
        if self.topic:      # nci_scores: the beam scores, or ones for a single returned sequence (main_models.py:3678-3682)
            weights = torch.ones((B, 1), dtype=torch.float32) if R == 1 else torch.tensor(scores, dtype=torch.float32)
        beam_recon = None
        if self.topic and self.C > 1 and self.ratio:       # reconstruct vectors of the beam code paths (pq.get_reconstruct_vector)
            beam_recon = self.pq.get_reconstruct_vector(torch.from_numpy(codes.reshape(B * R, self.M)).to(self.dev)).contiguous()
        ranked, ndoc = self.fine.rerank(qemb, codes, aggregate=self.aggregate, beam_weights=weights,
                                        doc_proba=self.doc_proba if self.topic else None, ratio=self.ratio if self.topic else 0.0,
                                        beam_recon=beam_recon)
        if getattr(a, "knn_topk_by_step", 0):       # main_models.py:3919-3995: a running torch.topk over the cluster chunks
            pool = max(a.recall_num)                  # keeps the pool_size best of a query's candidates, best first
            ranked = [(d[:pool], s_[:pool]) for d, s_ in ranked]
        nq = self.nq
        gt_s = self.fine.gt_scores(qemb, doc_ids) if self.hn_log is not None and nq is None else None
        results = []
        for i, text in enumerate(texts):
            d = codes[i].tolist()
            docs, sc = ranked[i]
            if want_c:
                cr, gt_codes = self._coarse_ranks(d, doc_ids[i])
                self.coarse_log.add((text, d, gt_codes, scores[i].tolist()) if nq is None else (text, d, scores[i].tolist()))
            if nq is None:
                self.fine_log.add((text, docs.tolist(), doc_ids[i]))
                fr = mfine.fine_ranks(docs, doc_ids[i])
            else:      # main_models.py:4060-4077: first ranked doc answering the question
                self.fine_log.add((text, docs.tolist()))
                fr = [nq.first_hit(doc_ids[i], docs)]
            if self.hn_log is not None:
                n = a.save_hard_neg
                self.hn_log.add((text, mfine.f32_repr(gt_s[i]) if nq is None else "", join_i64(docs[:n]),
                                 mfine.f32_repr(sc[:n])))
            results.append((text, int(ndoc[i]), cr, fr) if want_c else (text, int(ndoc[i]), fr))
        return self._timed(results, t1 if timing else None)

    def _coarse_ranks(self, d, gts):
        """Ranks of a sample's gt clusters among its beam clusters `d` (+ the gt codes the coarse log prints)."""
        if self.nq is not None:   # main_models.py:3738-3757: first beam cluster holding a document that answers the question
            answering = self.nq.docs_answering(gts)
            return [next((j for j, c in enumerate(d) if np.isin(self.index.lookup(c), answering).any()), None)], None
        if self.C > 1:            # use_pq_topk_label (main_models.py:3761-3771): best rank over a gt doc's C paths
            gt_codes = [self.doc_topk[g].tolist() for g in gts]
            return tuple(min((d.index(g) for g in paths if g in d), default=None) for paths in gt_codes), gt_codes
        gt_codes = [list(self.mapping[g]) for g in gts]
        return tuple(d.index(g) if g in d else None for g in gt_codes), gt_codes

    def _timed(self, results, t1):
        """--timing_infer_step: close the step's 'knn' clock; after N + 1 steps dump and exit as the reference."""
        if t1 is not None:
            self.timer["knn"].append(time.time() - t1)
            if self.timing_step_for_infer >= self.args.timing_infer_step:
                with open(f"times{self.R}.pkl", "wb") as f:
                    pickle.dump(self.timer, f)
                raise SystemExit(0)                     # the reference exit()s here, logs unmerged
            self.timing_step_for_infer += 1
        return results

    def _start_corpus_upload(self, a):
        """(thread, width, result dict) of a background `upload_rows` of the corpus file, or None (see __init__)."""
        import threading

        d = _corpus_width_hint(a)
        if d is None:
            return None
        n_docs = os.path.getsize(a.embedding_path) // (4 * d)
        res = {}
        dev = self.dev

        def work():
            try:
                torch.cuda.set_device(dev)
                emb = np.memmap(a.embedding_path, dtype=np.float32, mode="r", shape=(n_docs, d))
                with torch.cuda.stream(torch.cuda.Stream(device=dev)):       # its own stream: the model build's kernels do not queue behind 64 MiB copies
                    res["emb"] = upload_rows(emb, dev)
                    torch.cuda.current_stream().synchronize()
            except BaseException as e:      # re-raised by the joining thread
                res["error"] = e

        th = threading.Thread(target=work, name="mevi-corpus-upload", daemon=True)
        th.start()
        return th, d, res

    def run(self, df):
        a = self.args
        idx = rank_slice(len(df), self.rank, self.nrank)
        cache = []
        # --eval_batch_size is 2 in the reference's scripts (its infer() is per-sample Python); every output here is
        # per query and independent of how queries are grouped (row-wise kernels, tested), so the GPU is fed
        # device_batch_size queries at a time and the logs keep the sampler's order.
        bs = max(1, a.eval_batch_size, getattr(a, "device_batch_size", None) or 1)
        if self.timer is not None:      # latency of the script's own step: eval_batch_size queries, replayed HIP graphs
            bs = max(1, a.eval_batch_size)
        # the prefix tables of the PAWA head are sized for THIS run: a position is tabled only when this rank's queries x beams
        # outnumber its prefixes (nci.PrefixTables.WORTH_MARGIN) -- a one-shot eval does not pay seconds of table build for
        # prefixes it never visits.  MEVI_PREFIX_TABLES=all keeps the long-lived-model policy (every table that fits).
        if not self.eval_all and os.environ.get("MEVI_PREFIX_TABLES", "workload") != "all":
            self.nci.expect_queries(len(idx))
        for s in range(0, len(idx), bs):
            rows = df.iloc[idx[s:s + bs]]
            cache += self.infer(rows["query"].tolist(), rows["oldid"].tolist(), idx[s:s + bs])
        return self.finish(cache)

    # ---- handle_infer_results / validation_epoch_end ----------------------------------------
    def finish(self, cache):
        a = self.args
        if self.coarse_log is not None:
            self.coarse_log.merge()
        if self.fine_log is not None:
            self.fine_log.merge()
        part = f"/tmp/{os.path.basename(a.custom_save_path)}.results_{self.rank}"
        with open(part, "wb") as f:
            pickle.dump(cache, f)
        self.barrier()
        if self.hn_log is not None:
            self.hn_log.merge()
        out = None
        if self.rank == 0:
            allres = []
            for r in range(self.nrank):
                p = f"/tmp/{os.path.basename(a.custom_save_path)}.results_{r}"
                with open(p, "rb") as f:
                    allres += pickle.load(f)
                os.remove(p)
            out = summarize(allres, a.recall_num, self.R, both=self.level == "both", at_all=self.level == "fine")
            write_metrics(out, a.metric_path, self.R, len(self.index.keys))
        self.barrier()
        return out


def _acc(v, tables):
    found = [x for x in v if x is not None]
    best = min(found) if found else None
    for recall, mrr, hit in [tables]:
        for k in recall:
            if found:
                recall[k] += sum(x < k for x in found) / len(v)
                mrr[k] += 1 / (best + 1) if best < k else 0
                hit[k] += best < k
    return found, best


def summarize(results, recall_num, R, both=True, at_all=None):
    """recall / mrr / hitrate at recall_num for the ranks the samples carry (fine ranks; cluster ranks with recall_level
    'coarse') and mean ndoc; with recall_level 'both' cluster_* at the cut-offs <= R, with 'fine' (`at_all`) the extra key
    f'cluster{R}' = found-at-all figures (handle_infer_results, main_models.py:4100-4201)."""
    at_all = (not both) if at_all is None else at_all
    if both:
        queries = {q: (length, findex, cindex) for (q, length, cindex, findex) in results}
    else:
        queries = {q: (length, findex, None) for (q, length, findex) in results}
    fine = tuple({k: 0 for k in recall_num} for _ in range(3))
    ccut = sorted(k for k in recall_num if k <= R)
    if not ccut or ccut[-1] != R:
        ccut.append(R)
    coarse = tuple({k: 0 for k in ccut} for _ in range(3))
    nsamples = 0
    found_at_all = [0, 0, 0]
    for q, (length, findex, cindex) in queries.items():
        found, best = _acc(findex, fine)
        if both:
            _acc(cindex, coarse)
        elif at_all:
            found_at_all[0] += len(found) / len(findex)
            found_at_all[1] += 1 / (best + 1) if best is not None else 0
            found_at_all[2] += len(found) > 0
        nsamples += length
    n = len(queries)
    if at_all and not both:
        for t, v in zip(fine, found_at_all):
            t[f"cluster{R}"] = v
    for t in fine + coarse:
        for k in t:
            t[k] /= n
    out = dict(recall=fine[0], mrr=fine[1], hitrate=fine[2], ndoc=nsamples / n, nqueries=n)
    if both:
        out.update(cluster_recall=coarse[0], cluster_mrr=coarse[1], cluster_hitrate=coarse[2])
    return out


def write_metrics(out, metric_path, R, npqclus):
    lines = []
    for name in ("recall", "mrr", "hitrate"):
        lines += [f"{name}{k} {v}" for k, v in out[name].items()]
    for name in ("cluster_recall", "cluster_hitrate"):
        if name in out:
            lines += [f"{name}{k} {v}" for k, v in out[name].items()]
    lines.append(f"ndocs@cluster{R}: {out['ndoc']}")
    print(f"npqclus: {npqclus}")
    print("\n".join(lines))
    if metric_path:
        os.makedirs(os.path.dirname(os.path.abspath(metric_path)), exist_ok=True)
        with open(metric_path, "w") as f:
            f.write("\n".join(lines) + "\n")


def default_metric_path(args):
    logs = args.logs_dir or os.path.join(args.data_dir, "logs")
    t = args.time_str or time.strftime("%Y%m%d%H%M%S")
    tag = f"mevi_{args.dataset}_{args.query_type}_{args.model_info}_k{2 ** args.subvector_bits}"
    return os.path.join(logs, f"{tag}_metrics_{t}.txt")
