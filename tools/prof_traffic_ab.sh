#!/bin/bash
# L2 / HBM-side traffic of ip_filter_h1_kernel for two source trees on the SAME box (one dense search each via probe_dense.py):
#   tools/prof_traffic_ab.sh <treeA> <treeB>     (a tree = a checkout with its library built)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for T in "$@"; do
  name=$(basename $T); OUT=$R/gpurun_out/traffic_$name; rm -rf $OUT; mkdir -p $OUT
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $T/tools/probe_dense.py 8841823 > $OUT/fetch.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- python3 $T/tools/probe_dense.py 8841823 > $OUT/tcc.log 2>&1
  echo "== $name"; grep "^v0" $OUT/tcc.log | tail -1
  python3 $R/tools/traffic_summary.py $OUT | grep "launches\|per_launch\|hit_rate"
done
