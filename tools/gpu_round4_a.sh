#!/bin/bash
# round 4, first pass: whole GPU suite (incl. the new self-launch and graph-survival tests), smoke, the bench line as the driver runs it
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4a
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r4a/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4a/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
MEVI_BENCH_DETAIL=$R/gpurun_out/r4a/bench_detail.json timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err; echo "bench rc=$?"; wc -c gpurun_out/r4a/bench.json; tail -c 3000 gpurun_out/r4a/bench.json
