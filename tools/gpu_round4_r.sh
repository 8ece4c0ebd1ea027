# passage attention (attention_h16_kernel) with all loads first: tests, bits A/B, passage tower A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_t5_gpu.py -x -q -m gpu 2>&1 | tail -3
for v in base work; do MEVI_PROBE_LIB=tools/probes/ab/lib$v.so python tools/probe_attn_passage.py 2>&1 | tail -4; done
for i in 1 2; do
for v in base work; do echo "== $v"; MEVI_PROBE_LIB=tools/probes/ab/lib$v.so python tools/bench_passage.py 8192 2048 2>&1 | tail -1; done
done
