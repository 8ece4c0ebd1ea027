"""A few lines out of a bench.py JSON line: python tools/bench_summary.py <file>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("value %.0f %s  ms/step %.2f  frac %.4f  %s  avg launch %.3f ms" % (
    d["value"], d["unit"], d["ms_per_step"], r.get("frac") or 0, r.get("kernel"), r.get("avg_launch_ms") or 0))
c = d.get("config", {})
print("chain_c4", c.get("chain_c4"))
print("mrr10_match", c.get("mrr10_match"))
s = d.get("seq2seq_arm", {})
print("nci q/s", s.get("nci_generate_queries_per_s"), "frac", (s.get("roofline") or {}).get("frac"), "tables", s.get("prefix_tables"))
print("tower q/s", d.get("dense_arm_with_tower", {}).get("tower_queries_per_s"))
print("3x256", {k: v for k, v in (d.get("seq2seq_arm_rq_3x256") or {}).items() if not isinstance(v, (dict, str)) or k == "prefix_tables"}
      if isinstance(d.get("seq2seq_arm_rq_3x256"), dict) else d.get("seq2seq_arm_rq_3x256"))
print("sweep", d.get("seq2seq_batch_sweep"))
print("index_build", d.get("index_build"))
print("small", d.get("dense_small_batch"))
print("cli", d.get("faiss_search_cli_inclusive"))
print("cpu", d.get("cpu_baseline"))
print({k: v for k, v in d.items() if k.endswith("_error")})
