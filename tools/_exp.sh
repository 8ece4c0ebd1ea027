timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_t5_gpu.py tests/test_e2e_gpu.py -x -q 2>&1 | tail -3
for v in before now before now; do echo "== $v"; MEVI_PROBE_LIB=tools/probes/ab/lib$v.so timeout 400 python tools/bench_stages.py 2>&1 | grep "nci gen"; done
