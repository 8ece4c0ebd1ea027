R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_t5_gpu.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
for v in base work; do echo "== $v"; MEVI_PROBE_LIB=tools/probes/ab/lib$v.so python tools/bench_nci.py 6980 6980 4 32 | tail -1; done
done
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/r4q/nci; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/bench_nci.py 6980 6980 4 32 > $OUT/log.txt 2>&1
python3 $R/tools/show_stats.py $OUT 16 | grep -i "few_keys\|mfma16\|total"
