#!/bin/bash
# ONE parametrised runner for the GPU passes (replaces the per-pass gpu_round{3,4}_*.sh scripts):
#     gpurun --timeout S -- bash tools/gpu_pass.sh <tag> <step> [<step> ...]
# Every step writes under gpurun_out/<tag>/ and prints a few summary lines.  Steps (arguments after a colon, comma separated):
#   suite[:pytest args]     pytest tests -m gpu -q  (default: whole GPU suite)
#   smoke                   __graft_entry__.smoke()
#   bench[:steps,warmup]    bench.py as the driver runs it (default 20,5) + a short summary of the line
#   stats:<name>:<script and args with '+' for spaces>     rocprofv3 --kernel-trace --stats of tools/<script> (kernel_stats csv kept)
#   tail:<name>:<script+args>   kernel trace of tools/<script> summarised over its timed pass only (tools/trace_tail.py)
#   stats_bench             the same of `bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-seq2seq-legs`
#   traffic                 PMC passes (FETCH_SIZE; TCC hit / miss) of the bench's filter + tools/traffic_summary.py
#   pmc:<name>:<counters '+'-separated>:<script+args>      one PMC pass (kernel trace only) of tools/<script>
#   py:<name>:<script+args> plain `python tools/<script> args` with its output kept in <name>.txt
#   env VAR=VALUE           export for the following steps
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
plus() { echo "$1" | tr '+' ' '; }
while [ $# -gt 0 ]; do
  step=$1; shift
  kind=${step%%:*}; rest=${step#*:}; [ "$rest" = "$step" ] && rest=""
  echo "=== $step"
  case $kind in
    env) export "$1"; shift;;
    suite)
      timeout 2400 python -m pytest $([ -z "$rest" ] && echo tests || plus "$rest") -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2;;
    bench)
      st=${rest%%,*}; wu=${rest#*,}; [ -z "$rest" ] && st=20 && wu=5
      MEVI_BENCH_DETAIL=$O/bench_detail.json timeout 1200 python bench.py --steps $st --warmup $wu > $O/bench.json 2> $O/bench.err
      echo "bench rc=$? bytes=$(wc -c < $O/bench.json)"; python3 tools/bench_summary.py $O/bench.json;;
    stats)
      name=${rest%%:*}; cmd=$(plus "${rest#*:}"); OUT=$O/stats_$name; rm -rf $OUT; mkdir -p $OUT
      (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/$cmd > $OUT/log.txt 2>&1)
      grep -v "^[EW]20" $OUT/log.txt | tail -n 3
      cp "$(find $OUT -name '*kernel_stats.csv' | head -1)" $O/${name}_kernel_stats.csv 2>/dev/null
      python3 tools/show_stats.py $OUT 12; find $OUT -name "*kernel_trace.csv" -delete;;
    tail)   # kernel trace of tools/<script>, summarised over its LAST phase only (TRACE_GAP=1: the timed pass; tools/trace_tail.py)
      name=${rest%%:*}; cmd=$(plus "${rest#*:}"); OUT=$O/trace_$name; rm -rf $OUT; mkdir -p $OUT
      (cd /tmp && TRACE_GAP=1 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/$cmd > $OUT/log.txt 2>&1)
      grep -v "^[EW]20" $OUT/log.txt | tail -n 2
      python3 tools/trace_tail.py $OUT $O/${name}_timed_pass_kernel_stats.csv > $O/${name}_timed_pass.txt; head -40 $O/${name}_timed_pass.txt; rm -rf $OUT;;
    stats_bench)
      OUT=$O/stats_bench; rm -rf $OUT; mkdir -p $OUT
      (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-seq2seq-legs > $OUT/bench.log 2>&1)
      cp "$(find $OUT -name '*kernel_stats.csv' | head -1)" $O/bench_kernel_stats.csv; tail -n 1 $OUT/bench.log > $O/bench_under_rocprof.json
      python3 tools/show_stats.py $OUT 6; find $OUT -name "*kernel_trace.csv" -delete;;
    traffic)
      T=$O/traffic; mkdir -p $T
      (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $T/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-seq2seq-legs > $T/fetch.log 2>&1); echo "fetch pass rc=$?"
      (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $T/tcc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-seq2seq-legs > $T/tcc.log 2>&1); echo "tcc pass rc=$?"
      python3 tools/traffic_summary.py $T > $O/filter_traffic.json; cat $O/filter_traffic.json; rm -rf $T/fetch $T/tcc;;
    pmc)
      name=${rest%%:*}; r2=${rest#*:}; ctr=$(plus "${r2%%:*}"); cmd=$(plus "${r2#*:}"); OUT=$O/pmc_$name; rm -rf $OUT; mkdir -p $OUT
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT -- python3 $R/tools/$cmd > $OUT/log.txt 2>&1); echo "pmc rc=$?"
      python3 tools/pmc_summary.py $OUT 2>&1 | tail -n 30;;
    py)
      name=${rest%%:*}; cmd=$(plus "${rest#*:}")
      timeout 1500 python tools/$cmd > $O/$name.txt 2>&1; echo "rc=$?"; grep -v "^[EW]20" $O/$name.txt | tail -n 40;;
    *) echo "unknown step $step";;
  esac
done
du -sh $R/gpurun_out | tail -1
