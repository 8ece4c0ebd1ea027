#!/bin/bash
# split-f16 passage attention: tests, then A/B of the passage tower
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4i
timeout 1500 python -m pytest tests/test_t5_gpu.py tests/test_ops_gpu.py tests/test_gemm_split_gpu.py -m gpu -x -q > gpurun_out/r4i/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4i/pytest.log
for rep in 1 2; do
for mode in f32 h16; do
  MEVI_ATTN_PASSAGE=$mode timeout 300 python tools/bench_passage.py 4096 2048 2>&1 | tail -1 | sed "s/^/$mode rep $rep: /"
done
done
