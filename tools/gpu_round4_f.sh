#!/bin/bash
# round 4: the 16x16x32 split GEMM -- parity tests, then A/B (MEVI_GEMM_MFMA=32 vs default) of the GEMM, NCI generate, the tower
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4f
timeout 1500 python -m pytest tests/test_gemm_split_gpu.py tests/test_t5_gpu.py tests/test_ops_gpu.py -m gpu -x -q > gpurun_out/r4f/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4f/pytest.log
for rep in 1 2; do
for shape in 32 16; do
  echo "== shape $shape rep $rep"
  MEVI_GEMM_MFMA=$shape timeout 300 python tools/bench_gemm_split.py 76906 2304 768 76906 768 768 76906 3072 768 76906 768 3072 69800 768 768 2>&1 | tail -5
  MEVI_GEMM_MFMA=$shape timeout 300 python tools/bench_nci.py 6980 8192 2>&1 | tail -1
  MEVI_GEMM_MFMA=$shape timeout 300 python tools/bench_tower.py 6980 2>&1 | tail -1
done
done
