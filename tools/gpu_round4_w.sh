R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests/test_t5_gpu.py tests/test_e2e_gpu.py -x -q -m gpu 2>&1 | tail -3
python - <<'P'
import torch, json, bench
bench.late_imports() if hasattr(bench, "late_imports") else None
P
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/w_bench.json 2> gpurun_out/w_bench.err; python - <<'P'
import json
d=json.loads(open("gpurun_out/w_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["index_build"]["passage_tower"], d["dense_arm_with_tower"])
P
