#!/usr/bin/env python3
"""CLI-level wall clock of the drop-in at MS MARCO size: the four scripts of marco_eval_nci_rq.sh /
marco_ensemble.sh as FRESH processes on synthetic files of the real sizes --

    generate.py --gen_query          6980 queries  -> query_emb.bin                       (MEVI/generate.py:283-311)
    faiss_search.py --param Flat     27 GB docemb.bin x 6980 queries -> dense TSV (172 MB) (MEVI/faiss_search.py:80-98)
    main.py --mode eval              every flag of MEVI/marco_eval_nci_rq.sh; run TWICE: first use (RQ encode of the corpus +
                                     the cluster / mapping pickles written) and with the pickles present (the scripts' state:
                                     marco_generate_embedding_n_rq.sh wrote them offline)
    ensemble_marco.py                every flag of MEVI/marco_ensemble.sh

-- with a REAL SentencePiece tokenizer (tests/spm_fixture.py), t5-base-shaped random checkpoints and the bench's corpus.
Every process runs with MEVI_PHASE_LOG (mevi_amd/phases.py): the per-phase lines say where a user's wait goes (interpreter
start-up + imports, checkpoint reads, the 27 GB upload, kernels, file writes).
  python tools/e2e_cli.py [scratch_dir] [n_docs] [out.json]"""
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import synth  # noqa: E402

scratch = sys.argv[1] if len(sys.argv) > 1 else "/tmp/mevi_e2e_cli"
N = int(sys.argv[2]) if len(sys.argv) > 2 else bench.N_DOCS
out_json = sys.argv[3] if len(sys.argv) > 3 else None
nq, M, K, d = bench.N_QUERIES, 4, 32, 768
dev = torch.device("cuda:0")
T0 = time.time()
record = {"docs": N, "queries": nq, "processes": []}


def say(name, secs):
    print(f"{name:78s} {secs:8.2f} s", flush=True)


# ---- synthetic files of the real sizes (untimed set-up) ---------------------------------------------------------------------
shutil.rmtree(scratch, ignore_errors=True)
for sub in ("ckpts/t5-ance", "origin", "ance", "logs"):
    os.makedirs(os.path.join(scratch, sub))
t = time.time()
from spm_fixture import build_t5_tokenizer_dir  # noqa: E402

build_t5_tokenizer_dir(os.path.join(scratch, "ckpts/t5-ance"))
W, TW, g, rn = synth.weights(dev, M, K)
torch.save({"state_dict": {"model." + k: v.cpu() for k, v in W.items()}}, os.path.join(scratch, "ckpts/nci.ckpt"))
torch.save({k: v.cpu() for k, v in TW.items()}, os.path.join(scratch, "ckpts/t5-ance/pytorch_model.bin"))
json.dump(dict(d_model=d, d_ff=3072, num_heads=12, d_kv=64, num_layers=12, num_decoder_layers=12, model_type="t5"),
          open(os.path.join(scratch, "ckpts/t5-ance/config.json"), "w"))
torch.save(torch.nn.Parameter(torch.stack([rn(K, d, s=0.05 / (1 + j)) for j in range(M)]).cpu()),
           os.path.join(scratch, "ance/rqcodebook4_5.pt"))
del W, TW
with open(os.path.join(scratch, "ance/docemb.bin"), "wb") as f:
    for a in range(0, N, 1 << 20):
        f.write(bench.gen_shard(a, min(a + (1 << 20), N), dev, N).cpu().numpy().tobytes())
rng = np.random.default_rng(0)
qfile = os.path.join(scratch, "origin/dev_mevi_dedup.tsv")
with open(qfile, "w") as f:
    for i in range(nq):
        words = " ".join(f"w{rng.integers(0, 300)}" for _ in range(int(np.clip(rng.poisson(7) + 1, 2, 20))))
        f.write(f"{words} q{i % 30} {i}\t{int(rng.integers(0, N))}\n")
torch.cuda.empty_cache()
say("set-up (untimed): checkpoints, tokenizer, docemb.bin, query file", time.time() - t)
del g, rn

env = dict(os.environ, PYTHONPATH=ROOT, MEVI_PHASE_LOG=os.path.join(scratch, "phases.log"))
A = lambda *p: os.path.join(scratch, *p)      # noqa: E731


def run(name, argv):
    open(env["MEVI_PHASE_LOG"], "w").close()
    t0 = time.time()
    r = subprocess.run([sys.executable, *argv], capture_output=True, text=True, env=env, cwd=ROOT)
    wall = time.time() - t0
    if r.returncode != 0:
        print(r.stdout[-1500:], r.stderr[-3000:], flush=True)
        raise SystemExit(f"{name} failed with rc {r.returncode}")
    phases = [l.rstrip("\n").split("\t") for l in open(env["MEVI_PHASE_LOG"]) if l.strip()]
    say(name + "   [fresh process]", wall)
    ph, accounted = [], 0.0
    for _, pname, secs, _ in phases:
        print(f"      {pname:72s} {float(secs):8.2f} s", flush=True)
        ph.append({"phase": pname, "s": float(secs)})
        accounted += float(secs)
    if phases:
        print(f"      {'(exit: teardown, interpreter shutdown)':72s} {max(wall - accounted, 0.0):8.2f} s", flush=True)
    record["processes"].append({"name": name, "wall_s": round(wall, 3), "phases": ph})
    return wall, r


total = 0.0
w, _ = run("generate.py --gen_query", [os.path.join(ROOT, "generate.py"), "--query_file", qfile, "--model_path", A("ckpts/t5-ance"),
                                       "--tokenizer_path", A("ckpts/t5-ance"), "--query_embedding_path", A("ance/query_emb.bin"),
                                       "--gpus", "0", "--gen_query"])
total += w
w, _ = run("faiss_search.py --param Flat", [os.path.join(ROOT, "faiss_search.py"), "--query_path", A("ance/query_emb.bin"),
                                            "--doc_path", A("ance/docemb.bin"), "--output_path", A("ance/dense.txt"),
                                            "--raw_query_path", qfile, "--param", "Flat"])
total += w
main_argv = [os.path.join(ROOT, "main.py"), "--n_gpu", "1", "--mode", "eval", "--query_type", "gtq", "--co_neg_from", "clus",
             "--model_info", "base", "--id_class", "bert_k30_c30_1", "--dataset", "marco", "--Rdrop", "0.", "--eval_batch_size", "2",
             "--encode_batch_size", "1024", "--document_encoder", "ance", "--recall_level", "both", "--qtower", "encmask_dec",
             "--query_embed_accum", "attenpool", "--pq_loss", "ce", "--pq_type", "rq", "--codebook", "1", "--subvector_num", "4",
             "--subvector_bits", "5", "--use_gumbel_softmax", "0", "--pq_softmax_tau", "1", "--pq_hard_softmax_topk", "1",
             "--pq_negative", "none", "--fixnci", "--fixpq", "--document_encoder_from_pretrained", "0", "--not_load_document_encoder", "0",
             "--no_nci_loss", "1", "--query_encoder", "twin", "--num_return_sequences", "10", "--save_hard_neg", str(N),
             "--pq_path", A("ance/rqcodebook4_5.pt"), "--pq_cluster_path", A("ance/rqclus4_5.pkl"), "--nci_ckpt", A("ckpts/nci.ckpt"),
             "--data_dir", A("origin"), "--newid_dir", A("ance"), "--document_path", A("ance/all_document"), "--ckpt_dir", A("ckpts"),
             "--embedding_path", A("ance/docemb.bin"), "--custom_save_path", A("ance/nci_result_rq45_top10.tsv"),
             "--logs_dir", A("logs")]
w_first, _ = run("main.py --mode eval (first use: RQ encode + pickles written)", main_argv)
outs = {}
for suffix in ("_coarse.tsv", "_fine.tsv", f"_hn{N}.tsv"):
    outs[suffix] = open(A("ance/nci_result_rq45_top10" + suffix), "rb").read()
w, _ = run("main.py --mode eval (cluster pickles present: the scripts' state)", main_argv)
total += w
same = all(open(A("ance/nci_result_rq45_top10" + s_), "rb").read() == b for s_, b in outs.items())
print(f"   second run's logs byte-identical to the first run's: {same}", flush=True)
record["main_py_runs_byte_identical"] = same
# queries whose beam clusters are all empty log an empty list, which the consumer rejects as the reference's eval_list('') does
# (ensemble_marco.py:85-89): a trained model does not emit such beams, the random one does for some -- give those the first
# dense hit so that the consumer runs (untimed patch of a synthetic artefact)
hn_path = A(f"ance/nci_result_rq45_top10_hn{N}.tsv")
dense_first = {}
with open(A("ance/dense.txt")) as f:
    for line in f:
        it = line.rstrip("\n").split("\t")
        dense_first[it[0]] = (it[2].split(",", 1)[0], it[3].split(",", 1)[0])
lines = open(hn_path).read().split("\n")
fixed = 0
for j, line in enumerate(lines):
    f_ = line.split("\t")
    if len(f_) == 4 and f_[2] == "":
        f_[2], f_[3] = dense_first[f_[0]]
        lines[j] = "\t".join(f_)
        fixed += 1
open(hn_path, "w").write("\n".join(lines))
print(f"   {fixed} queries with no fine candidate patched (random NCI weights)", flush=True)
w, r = run("ensemble_marco.py", [os.path.join(ROOT, "ensemble_marco.py"), "--mapping_file", A("ance/rqmapping4_5.pkl"), "--gt_file", qfile,
                                 "--ance_file", A("ance/dense.txt"), "--coarse_file", A("ance/nci_result_rq45_top10_coarse.tsv"),
                                 "--fine_file", hn_path, "--ofile", A("ance/ensemble_result.txt")])
total += w
say("CLI CHAIN TOTAL (generate + faiss_search + main.py with pickles + ensemble)", total)
say("  ... with main.py's first use instead (RQ encode + pickles written)", total - record["processes"][3]["wall_s"] + w_first)
record["cli_chain_wall_s"] = round(total, 2)
record["cli_chain_wall_s_first_use"] = round(total - record["processes"][3]["wall_s"] + w_first, 2)
record["queries_per_s_cli_chain"] = round(nq / total, 1)
print("   ensemble stdout tail:", r.stdout.strip().split("\n")[-1][:200])
for fn in sorted(os.listdir(A("ance"))):
    print("   ", fn, os.path.getsize(A("ance", fn)) // (1 << 20), "MiB")
if out_json:
    os.makedirs(os.path.dirname(os.path.abspath(out_json)), exist_ok=True)
    with open(out_json, "w") as f:
        json.dump(record, f, indent=1)
shutil.rmtree(scratch, ignore_errors=True)
