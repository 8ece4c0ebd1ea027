#!/bin/bash
# round 4: the 16x16x32 filter kernel -- dense parity tests, then the dense bench A/B (32x32x16 vs 16x16x32) on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4c
timeout 1500 python -m pytest tests/test_dense_gpu.py tests/test_ip_rank_gpu.py tests/test_c1_gpu.py tests/test_full_size_gpu.py -m gpu -x -q -k "not two_ranks and not rccl and not own_ranks" > gpurun_out/r4c/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4c/pytest.log
for rep in 1 2; do
for shape in 32 16; do
  MEVI_IP_FILTER_MFMA=$shape MEVI_BENCH_DETAIL=$R/gpurun_out/r4c/detail_$shape.json timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-seq2seq-legs > gpurun_out/r4c/bench_${shape}_$rep.json 2> gpurun_out/r4c/bench_$shape.err
  python - <<P
import json
d=json.load(open("gpurun_out/r4c/bench_${shape}_$rep.json"))
print("shape $shape rep $rep:", round(d["value"]), "q/s", round(d["ms_per_step"],2), "ms/step; filter", round(d["roofline"]["avg_launch_ms"],3), "ms/launch, frac", round(d["roofline"]["frac"],4), "fallback", d["roofline"]["queries_sent_to_exact_fallback"], "planted", d["config"]["planted_top1_ok"])
P
done
done
