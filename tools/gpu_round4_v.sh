R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_t5_gpu.py -x -q -m gpu 2>&1 | tail -5
for i in 1 2; do
python tools/bench_nci.py 6980 6980 4 32 | tail -1
MEVI_HEAD_FUSED=0 python tools/bench_nci.py 6980 6980 4 32 | tail -1
done
python tools/bench_nci.py 6980 6980 3 256 | tail -1
MEVI_HEAD_FUSED=0 python tools/bench_nci.py 6980 6980 3 256 | tail -1
