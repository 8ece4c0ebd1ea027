#!/bin/bash
# round 3, first GPU pass: test suite, shard simulator with the schedule sweep, the new bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r3a
python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3a/pytest.log
SWEEP=1 REPS=3 python tools/bench_shard_sim.py gpurun_out/r3a/shard_sim_sweep.json > gpurun_out/r3a/shard_sim.log 2>&1; echo "sim rc=$?"
grep -v "^[EW]2026" gpurun_out/r3a/shard_sim.log | tail -45
python bench.py --steps 3 --warmup 1 > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r3a/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3a/bench.json') if l.startswith('{')][-1])
for k,v in d.items():
    s=json.dumps(v)
    print(k, s[:600])
PY
