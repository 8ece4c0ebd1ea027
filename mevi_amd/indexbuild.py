"""Offline index build = `marco_generate_embedding_n_rq.sh` (`main.py --mode train --only_gen_rq 1 ...`): passage
embeddings, the RQ codebook and the cluster pickles, produced natively so that the eval path's inputs need not come
from the reference.  Mirrors on_validation_epoch_start of the reference up to its `exit()`
(MEVI/main_models.py:4236-4266):

  1. gen_doc_embedding (:3077-3180)   docemb.bin, unless it exists: every rank encodes rows // nrank passages (the last
                                      rank takes the rest) into `<prefix>_<rank>.bin`, rank 0 concatenates
  2. pq.initialize (pq.py:440-486)    rqcodebook*.pt, unless it exists: residual k-means on the embeddings (rank 0)
  3. gen_pq_doc_cluster (:3182-3220)  rqclus*.pkl / rqmapping*.pkl, unless they exist

Host side only: the device work is TwinTower / BertTower, mevi_amd.rq (k-means, RQ encode)."""
import os
import pickle

import numpy as np
import torch

from .rq import ProductQuantization
from .t5 import DEVICE_PASS_TOKENS


def doc_rank_range(n, rank, nrank):
    """n // nrank rows each, the last rank takes the remainder (main_models.py:3092-3098, generate.py:141-147)."""
    per = n // nrank
    return per * rank, n if rank + 1 == nrank else per * (rank + 1)


def embed_documents(encoder, tokens, masks, output_path, rank=0, nrank=1, barrier=None, batch_size=1024, tmp_prefix=None):
    """tokens / masks: i64 [N, L] arrays (memmaps); writes f32 [N, dim] to `output_path` through per-rank part files."""
    barrier = barrier or (lambda: None)
    n, length = tokens.shape
    dim = encoder.dim
    start, end = doc_rank_range(n, rank, nrank)
    prefix = tmp_prefix or output_path[:-4]
    part_path = f"{prefix}_{rank}.bin"
    part = np.memmap(part_path, dtype=np.float32, mode="w+", shape=(max(end - start, 1), dim))
    step = max(batch_size, DEVICE_PASS_TOKENS // length)      # results do not depend on the grouping
    for s in range(start, end, step):
        e = min(s + step, end)
        psg = {"input_ids": torch.from_numpy(np.array(tokens[s:e])), "attention_mask": torch.from_numpy(np.array(masks[s:e]))}
        part[s - start:e - start] = encoder.encode_passage(psg).cpu().numpy()
    part.flush()
    del part
    barrier()
    if rank == 0:
        allp = np.memmap(output_path, dtype=np.float32, mode="w+", shape=(n, dim))
        for r in range(nrank):
            a, b = doc_rank_range(n, r, nrank)
            if b > a:
                allp[a:b] = np.memmap(f"{prefix}_{r}.bin", dtype=np.float32, mode="r", shape=(b - a, dim))
        allp.flush()
        del allp
        for r in range(nrank):
            os.remove(f"{prefix}_{r}.bin")
    barrier()


def load_passage_tokens(args, tokenizer=None):
    """--document_path: either the prefix of `<prefix>_tokens.bin` / `<prefix>_masks.bin` (i64 [N, co_doc_length],
    prepare_passage_tokenized.py) or a `.tsv` of `id \\t title \\t content` tokenised here (main_models.py:1465-1491,
    3117-3131: 'Title: .. Text: ..' for T5-ANCE, `title [SEP] content` for the BERT towers)."""
    length = getattr(args, "co_doc_length", None) or 128
    if not args.document_path.endswith(".tsv"):
        tok = np.memmap(args.document_path + "_tokens.bin", dtype=np.int64, mode="r").reshape(-1, length)
        msk = np.memmap(args.document_path + "_masks.bin", dtype=np.int64, mode="r").reshape(-1, length)
        return tok, msk
    import pandas as pd

    docs = pd.read_csv(args.document_path, sep="\t", names=["odid", "title", "content"],
                       dtype={"odid": str, "title": str, "content": str})
    docs.fillna("", inplace=True)
    if tokenizer is None:
        from transformers import AutoTokenizer

        name = "t5-ance" if args.document_encoder == "ance" else "bert-base-uncased"
        tokenizer = AutoTokenizer.from_pretrained(os.path.join(args.ckpt_dir, name))
    if args.document_encoder == "ance":
        text = ("Title: " + docs["title"] + " Text: " + docs["content"]).tolist()
    else:
        text = (docs["title"] + tokenizer.sep_token + docs["content"]).tolist()
    from .io import encode_batch

    out = encode_batch(tokenizer, text, length, add_special_tokens=args.document_encoder in ("ance", "ar2"))
    return out["input_ids"].numpy(), out["attention_mask"].numpy()


def load_tower(args, device):
    from .evalrun import load_bert_tower, load_tower_weights
    from .t5 import TwinTower

    if args.document_encoder == "ance":
        tw, dims = load_tower_weights(os.path.join(args.ckpt_dir, "t5-ance"))
        return TwinTower(tw, dims=dims, device=device)
    if args.document_encoder == "cocondenser":
        return load_bert_tower(os.path.join(args.ckpt_dir, "co-condenser-marco-retriever"), device)
    name = "ar2g_marco_finetune.pkl" if getattr(args, "dataset", "marco") == "marco" else "ar2g_nq_finetune.pkl"
    return load_bert_tower(os.path.join(args.ckpt_dir, name), device)


def build_index(args, rank=0, nrank=1, barrier=None, device=None, tower=None, tokenizer=None):
    """Returns (n_docs, dim, number of clusters) on every rank."""
    barrier = barrier or (lambda: None)
    device = torch.device(device if device is not None else "cuda")
    map_path = args.pq_cluster_path.replace("clus", "mapping")
    if not os.path.exists(args.embedding_path):
        tower = tower or load_tower(args, device)
        tokens, masks = load_passage_tokens(args, tokenizer)
        print(f"Generate embedding from {doc_rank_range(len(tokens), rank, nrank)} in {len(tokens)} docs...")
        embed_documents(tower, tokens, masks, args.embedding_path, rank, nrank, barrier, args.encode_batch_size or 64)
        dim = tower.dim
        del tower
    else:
        dim = None
    pq_file = args.pq_path if args.pq_path and os.path.isfile(args.pq_path) else None
    if dim is None:
        if pq_file is not None:
            dim = int(torch.load(pq_file, map_location="cpu").shape[-1])
        else:
            tower = tower or load_tower(args, device)
            dim = tower.dim
            del tower
    n_docs = os.path.getsize(args.embedding_path) // (4 * dim)
    emb = np.memmap(args.embedding_path, dtype=np.float32, mode="r", shape=(n_docs, dim))
    pq = ProductQuantization("rq", args.subvector_num, args.subvector_bits, "l2", dim, device=device)
    need_clusters = not (os.path.exists(args.pq_cluster_path) and os.path.exists(map_path))
    doc_emb = None
    if rank == 0 and (pq_file is None or need_clusters):
        from .io import upload_rows

        doc_emb = upload_rows(emb, device)
    pq.initialize(args.pq_path, doc_emb, rank=rank, seed=getattr(args, "seed", 42) - 1)
    nclus = None
    if need_clusters:
        if rank == 0:
            # after a fresh training the reference groups by the k-means labels (get_document_cluster_simple,
            # main_models.py:3195-3203); encoding with the final codebook gives the assignment the eval path recomputes
            index = pq.get_document_cluster(doc_emb, 0, 1, as_index=True)
            cluster, mapping = index.to_dicts()
            with open(args.pq_cluster_path, "wb") as f:
                pickle.dump(cluster, f)
            with open(map_path, "wb") as f:
                pickle.dump(mapping, f)
            from .metrics import write_mapping_sidecar

            write_mapping_sidecar(map_path, index.doc_codes(n_docs))
            nclus = len(cluster)
        barrier()
    if nclus is None:
        with open(args.pq_cluster_path, "rb") as f:
            nclus = len(pickle.load(f))
    print("Number of all pq document clusters:", nclus)
    return n_docs, dim, nclus
