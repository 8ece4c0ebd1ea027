#!/bin/bash
# round 4 final checkpoint: whole GPU suite, bench as the driver runs it, rocprof stats of bench / NCI / tower / passage, PMC traffic of the filter
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s
cd $R; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
MEVI_BENCH_DETAIL=$O/bench_detail.json timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; wc -c $O/bench.json
python3 - <<'P'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r4s"
d=json.load(open(O+"/bench.json"))
print("value", round(d["value"]), "frac", round(d["roofline"]["frac"],4), "ms", round(d["ms_per_step"],2), d["roofline"]["kernel"])
print("chain", d["config"].get("chain_c4"))
print("nci", d.get("seq2seq_arm",{}).get("nci_generate_queries_per_s"), d.get("seq2seq_arm",{}).get("roofline",{}).get("frac"), "tower", d.get("dense_arm_with_tower",{}).get("tower_queries_per_s"))
print("3x256", d.get("seq2seq_arm_rq_3x256"))
print("index_build", d.get("index_build"))
print("small", d.get("dense_small_batch"))
print({k:v for k,v in d.items() if k.endswith("_error")})
P
cd /tmp && export TMPDIR=/tmp
OUT=$O/stats_bench; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-seq2seq-legs > $OUT/bench.log 2>&1
cp $(find $OUT -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv; tail -n 1 $OUT/bench.log > $O/bench_under_rocprof.json; python3 $R/tools/show_stats.py $OUT 6
find $OUT -name "*kernel_trace.csv" -delete
for what in nci nci3 tower passage; do
  OUT=$O/stats_$what; rm -rf $OUT; mkdir -p $OUT
  case $what in nci) args="tools/bench_nci.py 6980 8192";; nci3) args="tools/bench_nci.py 6980 8192 3 256";; tower) args="tools/bench_tower.py 6980";; passage) args="tools/bench_passage.py 4096 2048";; esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/$args > $OUT/log.txt 2>&1
  tail -n 1 $OUT/log.txt
  cp $(find $OUT -name "*kernel_stats.csv" | head -1) $O/${what}_kernel_stats.csv
  python3 $R/tools/show_stats.py $OUT 8
  find $OUT -name "*kernel_trace.csv" -delete
done
# HBM-side traffic of the filter (own PMC passes, kernel trace only)
T=$O/traffic; mkdir -p $T
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $T/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-seq2seq-legs > $T/fetch.log 2>&1; echo "fetch pass rc=$?"
timeout 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $T/tcc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-seq2seq-legs > $T/tcc.log 2>&1; echo "tcc pass rc=$?"
python3 $R/tools/traffic_summary.py $T > $O/filter_traffic.json; cat $O/filter_traffic.json
rm -rf $T/fetch $T/tcc
python $R/tools/bench_latency.py > $O/latency.txt 2>&1; tail -8 $O/latency.txt
du -sh $R/gpurun_out
