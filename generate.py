#!/usr/bin/env python3
"""Embedding CLI -- same argv and output file as the reference's MEVI/generate.py:283-311 (`--gen_query`):
raw little-endian f32 [n_queries, dim].  One process per GPU (`--gpus "0,1,..."`), contiguous query ranges
per rank (generate.py:74-82), per-rank files concatenated by rank 0.  The T5-ANCE tower runs on the HIP
kernels (mevi_amd.t5.TwinTower); tokenisation stays an HF call.

`gen_doc_embedding` mirrors MEVI/generate.py:116-187 (the passage side: pre-tokenised 128-token passages ->
docemb.bin).  The reference reaches it from Python only; here it also has a flag, `--gen_doc --document_dir D
--doc_embedding_path P` (an addition: SURVEY 8(f).1).
"""
import argparse
import os
import socket

import numpy as np
import pandas as pd

from mevi_amd.io import encode_batch, flush_rows, map_rows
from mevi_amd.phases import finish, mark


# padded tokens per device pass (= mevi_amd.t5.DEVICE_PASS_TOKENS); --batch_size only raises it: the embeddings do not
# depend on how rows are grouped, the reference's 128 rows per pass would leave the GPU mostly idle
DEVICE_PASS_TOKENS = 262144

# rank 0 concatenates the part files (27 GB for the corpus) while the others wait in a barrier: the reference's
# 24-hour collective timeout (generate.py:69,135), not the 10-minute default
import datetime  # noqa: E402

DIST_TIMEOUT = datetime.timedelta(hours=24)


def get_tokenizer(model_path):
    from mevi_amd.io import load_tokenizer

    return load_tokenizer(model_path)


def rank_range(n, rank, nrank):
    """[start, end) of `rank`: n // nrank rows each, the first n % nrank ranks take one more (generate.py:74-82)."""
    base, extra = divmod(n, nrank)
    start = base * rank + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def doc_rank_range(n, rank, nrank):
    """[start, end) of `rank` for documents: n // nrank each, the LAST rank takes the remainder (generate.py:141-147)."""
    per = n // nrank
    start = per * rank
    return start, (n if rank + 1 == nrank else start + per)


def load_document_encoder(model_path, ckpt_path, device):
    """T5-ANCE (tied T5 towers) or a BERT-family tower (coCondenser directory / AR2 `.pkl`), chosen like the
    reference's get_document_encoder + AutoModel (generate.py:31-44)."""
    import torch

    from mevi_amd.evalrun import load_bert_tower, load_tower_weights, tower_model_type
    from mevi_amd.t5 import TwinTower

    if tower_model_type(model_path) == "bert":
        assert ckpt_path is None, "--ckpt_path (fine-tuned T5 tower inside a training checkpoint) does not apply to BERT towers"
        return load_bert_tower(model_path, device)
    weights, dims = load_tower_weights(model_path)
    if ckpt_path is not None:  # fine-tuned tower inside a training checkpoint (generate.py:200-211)
        from mevi_amd.io import load_checkpoint

        sd = load_checkpoint(ckpt_path)
        sd = sd.get("state_dict", sd)
        pre = "document_encoder.lm_q."
        weights.update({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)})
    return TwinTower(weights, dims=dims, device=device)


def gen_query_embedding(rank, query_file, model_path, ckpt_path, tokenizer_path, output_path, batch_size, dim, gpus,
                        query_length=32, tokenizer=None, encoder=None):
    import torch
    import torch.distributed as dist

    nrank = len(gpus)
    if nrank > 1:
        dist.init_process_group(os.environ.get("MEVI_DIST_BACKEND", "nccl"), rank=rank, world_size=nrank,
                                timeout=DIST_TIMEOUT)
    device = torch.device(f"cuda:{gpus[rank]}")
    torch.cuda.set_device(device)
    mark("start-up + imports (torch, GPU context)")
    encoder = encoder or load_document_encoder(model_path, ckpt_path, device)
    mark("tower weights -> HBM", sync=True)
    tokenizer = tokenizer or get_tokenizer(tokenizer_path)
    df = pd.read_csv(query_file, names=["query", "oldid"], encoding="utf-8", header=None, sep="\t")["query"]
    mark("tokenizer + query file")
    start, end = rank_range(len(df), rank, nrank)
    cur_path = output_path[:-4] + f"_{rank}.bin" if nrank > 1 else output_path
    out = map_rows(cur_path, dim, "w+", rows=end - start)
    batch_size = max(batch_size, DEVICE_PASS_TOKENS // query_length)    # see DEVICE_PASS_TOKENS
    for s in range(start, end, batch_size):
        e = min(s + batch_size, end)
        tok = encode_batch(tokenizer, df[s:e], query_length)
        out[s - start:e - start] = encoder.encode_query(tok).cpu().numpy()
    flush_rows(out)
    mark("tokenise + tower + write embeddings")
    if nrank > 1:
        dist.barrier()
        if rank == 0:
            allq = map_rows(output_path, dim, "w+", rows=len(df))
            at = 0
            for r in range(nrank):
                part = map_rows(output_path[:-4] + f"_{r}.bin", dim)
                allq[at:at + part.shape[0]] = part
                at += part.shape[0]
            flush_rows(allq)
            for r in range(nrank):
                os.remove(output_path[:-4] + f"_{r}.bin")
        dist.barrier()
        dist.destroy_process_group()


def gen_doc_embedding(rank, document_dir, model_path, ckpt_path, output_path, batch_size, dim, gpus, doc_length=128,
                      encoder=None):
    """all_document_tokens.bin / all_document_masks.bin (i64 [N, doc_length]) -> f32 [N, dim] at `output_path`
    (MEVI/generate.py:116-187: per-rank part files `<output>_<rank>.bin`, merged by rank 0)."""
    import torch
    import torch.distributed as dist

    nrank = len(gpus)
    if nrank > 1:
        dist.init_process_group(os.environ.get("MEVI_DIST_BACKEND", "nccl"), rank=rank, world_size=nrank,
                                timeout=DIST_TIMEOUT)
    device = torch.device(f"cuda:{gpus[rank]}")
    torch.cuda.set_device(device)
    encoder = encoder or load_document_encoder(model_path, ckpt_path, device)
    tokens = np.memmap(os.path.join(document_dir, "all_document_tokens.bin"), dtype=np.int64, mode="r").reshape(-1, doc_length)
    masks = np.memmap(os.path.join(document_dir, "all_document_masks.bin"), dtype=np.int64, mode="r").reshape(-1, doc_length)
    n = tokens.shape[0]
    start, end = doc_rank_range(n, rank, nrank)
    part_path = output_path[:-4] + f"_{rank}.bin"
    part = map_rows(part_path, dim, "w+", rows=end - start)
    batch_size = max(batch_size, DEVICE_PASS_TOKENS // doc_length)
    for s in range(start, end, batch_size):
        e = min(s + batch_size, end)
        psg = {"input_ids": torch.from_numpy(np.array(tokens[s:e])),
               "attention_mask": torch.from_numpy(np.array(masks[s:e]))}
        part[s - start:e - start] = encoder.encode_passage(psg).cpu().numpy()
    flush_rows(part)
    del part
    if nrank > 1:
        dist.barrier()
    if rank == 0:
        allp = map_rows(output_path, dim, "w+", rows=n)
        at = 0
        for r in range(nrank):
            p_ = map_rows(output_path[:-4] + f"_{r}.bin", dim)
            allp[at:at + p_.shape[0]] = p_
            at += p_.shape[0]
            del p_
        flush_rows(allp)
        del allp
        for r in range(nrank):
            os.remove(output_path[:-4] + f"_{r}.bin")
    if nrank > 1:
        dist.barrier()
        dist.destroy_process_group()


def profile_generate_query(query_file, model_path, ckpt_path, tokenizer_path, step_num, query_length=32, tokenizer=None,
                           encoder=None, out_path="timer.pkl"):
    """Per-query latency of tokenise + encode at batch size 1, pickled as a list of seconds to `timer.pkl`
    (MEVI/generate.py:245-281, `--timing_infer_step N`)."""
    import inspect
    import pickle
    from time import time

    import torch

    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    encoder = encoder or load_document_encoder(model_path, ckpt_path, device)
    tokenizer = tokenizer or get_tokenizer(tokenizer_path)
    df = pd.read_csv(query_file, names=["query", "oldid"], encoding="utf-8", header=None, sep="\t")["query"]
    graph = "graph" in inspect.signature(encoder.encode_query).parameters     # T5 towers: the step replays one HIP graph
    timer = []
    for start in range(0, step_num):
        batch = list(df[start:min(start + 1, step_num)])
        t0 = time()
        tok = encode_batch(tokenizer, batch, query_length)
        (encoder.encode_query(tok, graph=True) if graph else encoder.encode_query(tok)).cpu().numpy()
        timer.append(time() - t0)
    with open(out_path, "wb") as fw:
        pickle.dump(timer, fw)
    return timer


def _free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--query_file", type=str, default=None)
    parser.add_argument("--model_path", type=str, required=True)
    parser.add_argument("--tokenizer_path", type=str, required=True)
    parser.add_argument("--query_embedding_path", type=str, default=None)
    parser.add_argument("--ckpt_path", type=str, default=None)
    parser.add_argument("--batch_size", type=int, default=128)
    parser.add_argument("--dim", type=int, default=768)
    parser.add_argument("--gpus", type=str, default=None)
    parser.add_argument("--gen_query", action="store_true", default=False)
    parser.add_argument("--timing_infer_step", type=int, default=0)
    parser.add_argument("--gen_doc", action="store_true", default=False)       # additions (see the module docstring)
    parser.add_argument("--document_dir", type=str, default=None)
    parser.add_argument("--doc_embedding_path", type=str, default=None)
    parser.add_argument("--doc_length", type=int, default=128)
    args = parser.parse_args()
    gpus = [int(g) for g in args.gpus.split(",")] if args.gpus is not None else [0]
    if args.gen_doc:
        assert args.document_dir is not None and args.doc_embedding_path is not None, \
            "Need to specify source path and target path!"
        common = (args.document_dir, args.model_path, args.ckpt_path, args.doc_embedding_path, args.batch_size, args.dim,
                  gpus, args.doc_length)
        if len(gpus) > 1:
            import torch.multiprocessing as mp

            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ["MASTER_PORT"] = str(_free_port())
            mp.spawn(gen_doc_embedding, nprocs=len(gpus), args=common)
        else:
            gen_doc_embedding(0, *common)
        raise SystemExit(0)
    if not args.gen_query:
        if args.timing_infer_step > 0:
            profile_generate_query(args.query_file, args.model_path, args.ckpt_path, args.tokenizer_path,
                                   args.timing_infer_step)
        raise SystemExit(0)
    assert args.query_file is not None and args.query_embedding_path is not None, \
        "Need to specify source path and target path!"
    common = (args.query_file, args.model_path, args.ckpt_path, args.tokenizer_path, args.query_embedding_path,
              args.batch_size, args.dim, gpus)
    if len(gpus) > 1:
        import torch.multiprocessing as mp

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(_free_port())
        mp.spawn(gen_query_embedding, nprocs=len(gpus), args=common)
    else:
        gen_query_embedding(0, *common)
        finish()
