#!/bin/bash
# round 4 checkpoint: whole GPU suite, smoke, the bench line as the driver runs it, kernel stats of the bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4g
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r4g/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4g/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
MEVI_BENCH_DETAIL=$R/gpurun_out/r4g/bench_detail.json timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4g/bench.json 2> gpurun_out/r4g/bench.err; echo "bench rc=$?"; wc -c gpurun_out/r4g/bench.json; python3 - <<'P'
import json
d=json.load(open("gpurun_out/r4g/bench.json"))
print("value", round(d["value"]), "frac", round(d["roofline"]["frac"],4), "chain", d["config"].get("chain_c4"), "nci", d.get("seq2seq_arm",{}).get("nci_generate_queries_per_s"), d.get("seq2seq_arm",{}).get("roofline",{}).get("frac"))
print({k:(v if not isinstance(v,dict) else "...") for k,v in d.items() if k.endswith("_error")})
P
