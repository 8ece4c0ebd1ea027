R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in base work; do MEVI_PROBE_LIB=tools/probes/ab/lib$v.so python tools/probe_rows_bits.py 2>&1 | tail -1; done
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_gemm_split_gpu.py tests/test_t5_gpu.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
for v in base work; do echo "== $v"; MEVI_PROBE_LIB=tools/probes/ab/lib$v.so python tools/bench_nci.py 6980 6980 4 32 | tail -1; done
done
