#!/bin/bash
# round 3: the bench line (driver form) + kernel stats of the same command + PMC traffic passes + full GPU test suite
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r3f
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r3f/bench.json 2> gpurun_out/r3f/bench.err; echo "bench rc=$?"; tail -4 gpurun_out/r3f/bench.err
bash tools/prof_stats.sh > gpurun_out/r3f/prof_stats.log 2>&1; tail -3 gpurun_out/r3f/prof_stats.log | cut -c1-200
bash tools/prof_traffic.sh > gpurun_out/r3f/prof_traffic.log 2>&1; tail -3 gpurun_out/r3f/prof_traffic.log | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3f/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3f/pytest.log
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3f/bench.json') if l.startswith('{')][-1])
for k,v in d.items():
    print(k, json.dumps(v)[:420])
PY
