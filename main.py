#!/usr/bin/env python3
"""`python main.py --mode eval ...` -- the argv surface of the reference's MEVI/main.py for the eval
path that marco_eval_nci_rq.sh drives (MEVI/main.py:356-794, 267-337).  Every flag of that script is
accepted; the ones that configure training are parsed and ignored.  Only --mode eval with
--codebook 1 --pq_type rq --document_encoder ance|cocondenser|ar2 --query_encoder twin --recall_level both|coarse|fine is built
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
    p.add_argument("--not_load_document_encoder", type=int, default=0)
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
    p.add_argument("--use_topic_model", type=int, default=0)
    p.add_argument("--topic_score_ratio", type=float, default=0.)
    p.add_argument("--knn_topk_by_step", type=int, default=0)
    p.add_argument("--only_gen_rq", type=int, default=0)
    p.add_argument("--co_doc_length", type=int, default=128)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--timing_infer_step", type=int, default=0)
    args, rest = p.parse_known_args(argv)
    # the rest must be flags of the reference's own parser; those that change the evaluation must carry the built value
    ignored = []
    i = 0
    while i < len(rest):
        tok = rest[i]
        name = tok[2:].split("=", 1)[0] if tok.startswith("--") else None
        if name in PASS_THROUGH_SWITCHES:
            ignored.append((tok, None))
            i += 1
            continue
        if name not in PASS_THROUGH_VALUE_FLAGS:
            raise SystemExit(f"main.py: unrecognized argument {tok!r} (not a flag of the reference's main.py)")
        if "=" in tok:
            value, i = tok.split("=", 1)[1], i + 1
        elif i + 1 < len(rest) and not (rest[i + 1].startswith("--") and rest[i + 1][2:3].isalpha()):
            value, i = rest[i + 1], i + 2
        else:
            raise SystemExit(f"main.py: argument {tok}: expected one argument")
        allowed = EVAL_AFFECTING.get(name, ())
        if allowed and value not in allowed:
            raise SystemExit(f"main.py --mode eval: --{name} {value} changes what is evaluated and is not built "
                             f"(built: {' | '.join(allowed)})")
        ignored.append(("--" + name, value))
    args.ignored_flags = ignored
    args.recall_num = sorted(int(r) for r in args.recall_num.split(","))
    if not args.document_encoder or args.recall_level == "coarse":      # MEVI/main.py:750-752
        args.recall_num = [r for r in args.recall_num if r <= args.num_return_sequences]
    n = eval(args.n_gpu) if not args.n_gpu.isdigit() else int(args.n_gpu)  # int or list literal (MEVI/main.py:734-737)
    args.n_gpu = list(range(n)) if isinstance(n, int) else list(n)
    if args.ckpt_dir is None:
        args.ckpt_dir = os.path.join(args.data_dir, "../ckpts")
    if args.document_encoder and args.encode_batch_size is None:
        args.encode_batch_size = 64
    return args


# Flags of the reference's parser (MEVI/main.py:343-731) that this build does not read.  They are accepted -- the eval
# scripts pass many training hyper-parameters -- but nothing else is: a flag outside the reference's parser (a typo)
# ends the run, as argparse ends the reference's.
PASS_THROUGH_VALUE_FLAGS = frozenset((
    "wandb_token wandb_id output_dir model_name_or_path tokenizer_name_or_path freeze_encoder freeze_embeds "
    "weight_decay adam_epsilon warmup_steps num_train_epochs gradient_accumulation_steps "
    "resume_from_checkpoint n_val n_train early_stop_callback fp_16 opt_level max_grad_norm pretrain_encoder "
    "limit_val_batches softmax aug accelerator num_layers num_decoder_layers d_ff d_model num_heads num_cls "
    "decode_embedding output_vocab_size hierarchic_decode tie_word_embedding tie_decode_embedding gen_method "
    "random_gen label_length_cutoff check_val_every_n_epoch val_check_interval train_batch_size "
    "max_input_length inf_max_input_length max_output_length doc_length contrastive_variant learning_rate "
    "decoder_learning_rate document_encoder_learning_rate projection_learning_rate certain_epoch given_ckpt "
    "qenc_ckpt penc_ckpt load_encoder_only id_class ckpt_monitor monitor_name "
    "Rdrop dropout_rate Rdrop_only_decoder Rdrop_loss adaptor_decode adaptor_efficient test1000 position "
    "contrastive embedding_distillation weight_distillation hard_negative aug_query aug_query_type "
    "sample_neg_num query_tloss weight_tloss ranking_loss disc_loss input_dropout denoising multiple_decoder "
    "decoder_num loss_weight kary tree mapping_path cluster_path tree_path eval_train_data drop_data_rate "
    "num_sanity_val_steps timing_step save_top_k drop_last reserve_decoder decoder_integration tie_encoders "
    "document_encoder_from_pretrained negatives_x_sample alt_granularity alt_train nci_twin_train_ratio "
    "qtower query_embed_accum co_neg_num co_neg_from co_neg_file co_neg_clus_file simans_hyper_a "
    "simans_hyper_b co_loss_scale no_nci_loss no_twin_loss pq_update_method pq_update_after_eval "
    "pq_init_method pq_dist_mode pq_loss pq_twin_loss pq_runtime_label pq_runtime_update_cluster "
    "use_gumbel_softmax pq_softmax_tau pq_hard_softmax_topk topk_sequence pq_negative pq_negative_margin "
    "pq_negative_loss tie_nci_pq_centroid aug_topk_clus aug_find_topk_from aug_sample_topk "
    "reconstruct_for_embeddings centroid_update_loss centroid_loss_scale infer_reconstruct_vector "
    "align_clustering query_vq_label nci_twin_alt_epoch nci_vq_alt_epoch rq_topk_score multiclus_label "
    "cat_cluster_centroid cluster_position_topk cluster_position_embedding "
    "cluster_position_rank_reciprocal cluster_position_proj_style use_cluster_adaptor "
    "cluster_adaptor_decouple cluster_adaptor_trainable_token_embedding "
    "cluster_adaptor_trainable_position_embedding cluster_adaptor_head_num cluster_adaptor_layer_num use_ort "
    "use_deepspeed ads_info"
).split())
PASS_THROUGH_SWITCHES = frozenset((
    "split_data validation_release_traindataset no_validation fixnci fixdocenc fixncienc fixncit5 fixpq "
    "fixlmq fixlmp fixproj fp16_opt"
).split())
# ... of which these CHANGE what --mode eval computes (model structure, decoding, scoring ablations).  Only the value
# the shipped eval scripts run with (= the reference's default unless noted) is built; anything else is refused
# instead of being silently ignored.
EVAL_AFFECTING = {
    "fp_16": ("0",), "decode_embedding": ("2",), "hierarchic_decode": ("0",), "tie_word_embedding": ("0",),
    "tie_decode_embedding": ("1",), "adaptor_decode": ("1",), "adaptor_efficient": ("1",), "position": ("1",),
    "multiple_decoder": ("0",), "decoder_num": ("1",), "reserve_decoder": ("0",), "decoder_integration": ("series",),
    "softmax": ("0",), "tree": ("1",), "topk_sequence": ("0",), "test1000": ("0",), "gen_method": ("greedy",),
    "cat_cluster_centroid": ("0",),
    "cluster_position_topk": ("0",), "use_cluster_adaptor": ("0",), "infer_reconstruct_vector": ("0",),
    "rq_topk_score": ("prod",), "multiclus_label": ("top1",), "reconstruct_for_embeddings": ("0",),
    "tie_encoders": ("1",), "input_dropout": ("0", "1"), "denoising": ("0",), "load_encoder_only": ("0",),
    "pq_dist_mode": ("l2",), "use_ort": ("0",), "eval_train_data": ("0",),
    "drop_data_rate": ("0", "0.0", "0."),      # > 0 reads dev_mevi_dedup_drop<rate>.tsv (main_utils.py:263-264)
}


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
    if a.test_set != "dev":       # load_data_infer has no other branch (main_utils.py:238: the reference dies on df = None)
        raise SystemExit(f"main.py --mode eval: --test_set {a.test_set!r} is not built (only 'dev')")
    if a.dataset not in ("marco", "nq_dpr"):
        raise SystemExit(f"main.py --mode eval: --dataset {a.dataset!r} is not built (marco, nq_dpr)")
    need = dict(codebook=1, pq_type="rq", query_encoder="twin")
    if a.recall_level not in ("both", "coarse", "fine"):
        raise SystemExit(f"main.py --mode eval: --recall_level {a.recall_level!r} is not built (both | coarse | fine)")
    if a.doc_multiclus < 1 or (a.doc_multiclus > 1 and (a.eval_all_documents or a.knn_topk_by_step)):
        raise SystemExit("main.py --mode eval: --doc_multiclus C > 1 is built for the cluster re-ranking path only")
    if a.eval_all_documents:   # the brute-force ablation; same preconditions as the reference (MEVI/main.py:657-658)
        need.update(recall_level="fine", knn_topk_by_step=1)
    for k, v in need.items():
        if getattr(a, k) != v:
            raise SystemExit(f"main.py --mode eval: --{k} {getattr(a, k)!r} is not built (only {v!r}, as in marco_eval_nci_rq.sh)")
    for k in ("pq_path", "pq_cluster_path", "embedding_path", "custom_save_path"):
        if getattr(a, k) is None:
            raise SystemExit(f"main.py --mode eval: --{k} is required")
    if a.nci_ckpt is None and a.infer_ckpt is None:   # try_load_ckpt asserts one of them (MEVI/main.py:201)
        raise SystemExit("main.py --mode eval: --nci_ckpt or --infer_ckpt is required")
    if a.num_return_sequences > (2 ** a.subvector_bits) ** a.subvector_num:
        raise SystemExit("num_return_sequences exceeds the number of code paths (2**subvector_bits)**subvector_num: the "
                         "reference would return -1e9 placeholder hypotheses")


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
    from mevi_amd.phases import mark

    mark("start-up + imports (torch, GPU context)")
    run = EvalRun(args, rank=rank, nrank=nrank, barrier=barrier, device=torch.device("cuda", gpu))
    df = (load_nq_queries if args.dataset == "nq_dpr" else load_queries)(args.data_dir, args.n_test)
    print("Inference start...")
    run.run(df)
    mark("EvalRun.run: beam search + tower + fine stage + logs + metrics", sync=True)
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
    if len(args.n_gpu) == 1 and "RANK" not in os.environ:
        from mevi_amd.phases import finish

        finish()
