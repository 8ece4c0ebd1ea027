R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_t5_gpu.py -x -q -m gpu -k "device_sized or batch_invariant or poison or grouping" 2>&1 | tail -3
for i in 1 2; do python tools/bench_nci.py 6980 6980 3 256 | tail -1; done
python tools/bench_nci.py 6980 6980 3 256 64 | tail -1
