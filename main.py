#!/usr/bin/env python3
"""`python main.py --mode eval ...` -- the argv surface of the reference's MEVI/main.py for the eval
path that marco_eval_nci_rq.sh drives (MEVI/main.py:356-794, 267-337).  Every flag of that script is
accepted; the ones that configure training are parsed and ignored.  Only --mode eval with
--codebook 1 --pq_type rq --document_encoder ance|cocondenser|ar2 --query_encoder twin --recall_level both is built
(+ the brute-force ablation --eval_all_documents 1 --recall_level fine --knn_topk_by_step 1)
(the configuration of every shipped eval script); anything else raises.

One process per GPU: `--n_gpu N` spawns N ranks itself like the reference (queries split by rank,
rank-local TSVs merged by rank 0 between two barriers); torchrun launches are honoured too.
"""
import argparse
import datetime
import os
import socket
import sys
import time


def parsers_parser(argv=None):
    p = argparse.ArgumentParser()
    # flags that matter on the eval path (defaults = MEVI/main.py)
    p.add_argument("--mode", type=str, default="train", choices=["train", "eval", "calculate"])
    p.add_argument("--n_gpu", type=str, default="1")
    p.add_argument("--model_info", type=str, default="base", choices=["small", "large", "base", "3b", "11b"])
    p.add_argument("--dataset", type=str, default="marco")
    p.add_argument("--query_type", type=str, default="gtq")
    p.add_argument("--query_embedding_path", type=str, default=None,
                   help="(this build) query_emb.bin written by generate.py --gen_query for the same query file: the fine "
                        "stage reads it instead of running the query tower a second time (same bits)")
    p.add_argument("--eval_batch_size", type=int, default=2)
    p.add_argument("--encode_batch_size", type=int, default=None)
    p.add_argument("--device_batch_size", type=int, default=8192,
                   help="(this build) queries per GPU pass; results do not depend on it. --eval_batch_size only raises it")
    p.add_argument("--document_encoder", type=str, default=None)
    p.add_argument("--query_encoder", type=str, default="twin")
    p.add_argument("--recall_level", type=str, default="coarse")
    p.add_argument("--recall_num", type=str, default="1,5,10,20,50,100")
    p.add_argument("--codebook", type=int, default=0)
    p.add_argument("--pq_type", type=str, default="pq")
    p.add_argument("--subvector_num", type=int, default=32)
    p.add_argument("--subvector_bits", type=int, default=8)
    p.add_argument("--num_return_sequences", type=int, default=100)
    p.add_argument("--length_penalty", type=int, default=0.8)  # (sic) type=int, default 0.8: MEVI/main.py:405
    p.add_argument("--adaptor_layer_num", type=int, default=4)
    p.add_argument("--save_hard_neg", type=int, default=0)
    p.add_argument("--pq_path", type=str, default=None)
    p.add_argument("--pq_cluster_path", type=str, default=None)
    p.add_argument("--nci_ckpt", type=str, default=None)
    p.add_argument("--infer_ckpt", type=str, default=None)
    p.add_argument("--data_dir", type=str, required=True)
    p.add_argument("--newid_dir", type=str, default=None)
    p.add_argument("--document_path", type=str, default=None)
    p.add_argument("--ckpt_dir", type=str, default=None)
    p.add_argument("--embedding_path", type=str, default=None)
    p.add_argument("--custom_save_path", type=str, default=None)
    p.add_argument("--logs_dir", type=str, default=None)
    p.add_argument("--time_str", type=str, default=None)
    p.add_argument("--n_test", type=int, default=-1)
    p.add_argument("--test_set", type=str, default="dev")
    p.add_argument("--doc_multiclus", type=int, default=1)
    p.add_argument("--multiclus_score_aggr", type=str, default="add", choices=["add", "max"])
    p.add_argument("--eval_all_documents", type=int, default=0)
    p.add_argument("--knn_topk_by_step", type=int, default=0)
    p.add_argument("--only_gen_rq", type=int, default=0)
    p.add_argument("--co_doc_length", type=int, default=128)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--timing_infer_step", type=int, default=0)
    args, rest = p.parse_known_args(argv)
    # training / ablation flags of marco_eval_nci_rq.sh: accepted, no effect on eval
    ignored = []
    i = 0
    while i < len(rest):
        tok = rest[i]
        if not tok.startswith("--"):
            raise SystemExit(f"main.py: unexpected argument {tok!r}")
        if i + 1 < len(rest) and not rest[i + 1].startswith("--"):
            ignored.append((tok, rest[i + 1]))
            i += 2
        else:
            ignored.append((tok, None))
            i += 1
    args.ignored_flags = ignored
    args.recall_num = sorted(int(r) for r in args.recall_num.split(","))
    n = eval(args.n_gpu) if not args.n_gpu.isdigit() else int(args.n_gpu)  # int or list literal (MEVI/main.py:734-737)
    args.n_gpu = list(range(n)) if isinstance(n, int) else list(n)
    if args.ckpt_dir is None:
        args.ckpt_dir = os.path.join(args.data_dir, "../ckpts")
    if args.document_encoder and args.encode_batch_size is None:
        args.encode_batch_size = 64
    return args


def check_supported(a):
    if a.mode == "train" and a.only_gen_rq:   # marco_generate_embedding_n_rq.sh: embeddings + RQ codebook + clusters, then exit
        if a.document_encoder not in ("ance", "cocondenser", "ar2") or a.pq_type != "rq" or not a.codebook:
            raise SystemExit("main.py --only_gen_rq 1: needs --codebook 1 --pq_type rq --document_encoder ance|cocondenser|ar2")
        for k in ("pq_path", "pq_cluster_path", "embedding_path", "document_path", "ckpt_dir"):
            if getattr(a, k) is None:
                raise SystemExit(f"main.py --only_gen_rq 1: --{k} is required")
        return
    if a.mode != "eval":
        raise SystemExit("mevi_amd builds the inference hot path only: use --mode eval, or --mode train --only_gen_rq 1 "
                         "for the offline index build (training is out of scope)")
    if a.document_encoder not in ("ance", "cocondenser", "ar2"):
        raise SystemExit(f"main.py --mode eval: --document_encoder {a.document_encoder!r} is not built")
    if a.dataset not in ("marco", "nq_dpr"):
        raise SystemExit(f"main.py --mode eval: --dataset {a.dataset!r} is not built (marco, nq_dpr)")
    need = dict(codebook=1, pq_type="rq", query_encoder="twin", recall_level="both")
    if a.doc_multiclus < 1 or (a.doc_multiclus > 1 and (a.eval_all_documents or a.knn_topk_by_step)):
        raise SystemExit("main.py --mode eval: --doc_multiclus C > 1 is built for the cluster re-ranking path only")
    if a.eval_all_documents:   # the brute-force ablation; same preconditions as the reference (MEVI/main.py:657-658)
        need.update(recall_level="fine", knn_topk_by_step=1)
    for k, v in need.items():
        if getattr(a, k) != v:
            raise SystemExit(f"main.py --mode eval: --{k} {getattr(a, k)!r} is not built (only {v!r}, as in marco_eval_nci_rq.sh)")
    for k in ("nci_ckpt", "pq_path", "pq_cluster_path", "embedding_path", "custom_save_path"):
        if getattr(a, k) is None:
            raise SystemExit(f"main.py --mode eval: --{k} is required")
    if a.num_return_sequences > 2 ** a.subvector_bits:
        raise SystemExit("num_return_sequences > 2**subvector_bits is not pinned by the reference (SURVEY 8(a') note ii)")


def _free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def partial_inference(rank, args):
    import torch
    import torch.distributed as dist

    from mevi_amd.evalrun import EvalRun, default_metric_path, load_nq_queries, load_queries

    nrank = int(os.environ.get("WORLD_SIZE", len(args.n_gpu)))
    if "RANK" in os.environ:
        rank = int(os.environ["RANK"])
    gpu = args.n_gpu[rank] if rank < len(args.n_gpu) else int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(gpu)
    barrier = None
    if nrank > 1:
        # rank 0 alone trains the RQ codebook / pickles 8.8 M-entry dicts while the others wait: 24 h like the reference
        # (main.py:285-287), not the 10-minute default
        dist.init_process_group(os.environ.get("MEVI_DIST_BACKEND", "nccl"), rank=rank, world_size=nrank,
                                timeout=datetime.timedelta(hours=24))
        barrier = dist.barrier
    if args.mode == "train":      # --only_gen_rq 1 (check_supported)
        from mevi_amd.indexbuild import build_index

        build_index(args, rank=rank, nrank=nrank, barrier=barrier, device=torch.device("cuda", gpu))
        if nrank > 1:
            dist.destroy_process_group()
        return
    if args.time_str is None:
        args.time_str = time.strftime("%Y%m%d%H%M%S")
    args.metric_path = default_metric_path(args)
    run = EvalRun(args, rank=rank, nrank=nrank, barrier=barrier, device=torch.device("cuda", gpu))
    df = (load_nq_queries if args.dataset == "nq_dpr" else load_queries)(args.data_dir, args.n_test)
    print("Inference start...")
    run.run(df)
    if nrank > 1:
        dist.destroy_process_group()


def inference(args):
    nrank = len(args.n_gpu)
    if nrank > 1 and "RANK" not in os.environ:
        import torch.multiprocessing as mp

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(_free_port())
        mp.spawn(partial_inference, nprocs=nrank, args=(args,))
    else:
        partial_inference(0, args)


if __name__ == "__main__":
    args = parsers_parser()
    check_supported(args)
    if args.ignored_flags:
        print("accepted and ignored (training / ablation flags):", " ".join(f for f, _ in args.ignored_flags), file=sys.stderr)
    inference(args)
