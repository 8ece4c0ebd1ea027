#!/bin/bash
# per-kernel time of the query tower for library variants built by tools/ab_build.sh:  tools/prof_tower_ab.sh base vg2 ...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  OUT=$R/gpurun_out/tower_$v; rm -rf $OUT; mkdir -p $OUT
  export MEVI_PROBE_LIB=$R/tools/probes/ab/lib$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/bench_tower.py > $OUT/log.txt 2>&1
  echo "== $v: $(tail -n 1 $OUT/log.txt)"
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  grep -i "varlen\|mfma32\|attention_kernel" $f | cut -d, -f1-4 | cut -c1-150
done
