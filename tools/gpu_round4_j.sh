#!/bin/bash
# round 4 checkpoint 2: whole GPU suite, bench as the driver runs it, rocprof stats of bench / NCI / tower / passage
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4j
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r4j/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4j/pytest.log
MEVI_BENCH_DETAIL=$R/gpurun_out/r4j/bench_detail.json timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4j/bench.json 2> gpurun_out/r4j/bench.err; echo "bench rc=$?"; wc -c gpurun_out/r4j/bench.json
python3 - <<'P'
import json
d=json.load(open("gpurun_out/r4j/bench.json"))
print("value", round(d["value"]), "frac", round(d["roofline"]["frac"],4), "ms", round(d["ms_per_step"],2))
print("chain", d["config"].get("chain_c4"))
print("nci", d.get("seq2seq_arm",{}).get("nci_generate_queries_per_s"), d.get("seq2seq_arm",{}).get("roofline",{}).get("frac"), "tower", d.get("dense_arm_with_tower",{}).get("tower_queries_per_s"))
print("index_build", d.get("index_build"))
print("small", d.get("dense_small_batch"), "cli", d.get("faiss_search_cli_inclusive"))
print({k:v for k,v in d.items() if k.endswith("_error")})
P
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/r4j/stats_bench; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-seq2seq-legs > $OUT/bench.log 2>&1
cp $(find $OUT -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r4j/bench_kernel_stats.csv; python3 $R/tools/show_stats.py $OUT 6
for what in nci tower passage; do
  OUT=$R/gpurun_out/r4j/stats_$what; rm -rf $OUT; mkdir -p $OUT
  case $what in nci) args="tools/bench_nci.py 6980 8192";; tower) args="tools/bench_tower.py 6980";; passage) args="tools/bench_passage.py 4096 2048";; esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/$args > $OUT/log.txt 2>&1
  tail -n 1 $OUT/log.txt
  cp $(find $OUT -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r4j/${what}_kernel_stats.csv
  python3 $R/tools/show_stats.py $OUT 8
done
