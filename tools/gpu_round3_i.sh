#!/bin/bash
# round 3, final pass: whole GPU suite, smoke, the bench line as the driver runs it, kernel stats of the bench and of NCI generate
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r3i
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3i/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3i/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r3i/bench.json 2> gpurun_out/r3i/bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r3i/bench.json
bash tools/prof_stats.sh 2>&1 | tail -3
cp $(find gpurun_out/stats -name "*kernel_stats.csv" | head -1) gpurun_out/r3i/bench_kernel_stats.csv
bash tools/prof_nci.sh 6980 8192 2>&1 | tail -1
cp $(find gpurun_out/nci -name "*kernel_stats.csv" | head -1) gpurun_out/r3i/nci_kernel_stats.csv
