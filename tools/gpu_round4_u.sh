R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python tools/bench_latency.py 2>&1 | tail -14
