# stress of the XDIR RQ kernel's counted waits: the whole corpus ten times, fast codes against the exact kernel's each time
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in 1 2 3 4 5 6 7 8 9 10; do python tools/bench_rq.py 8841823 gpurun_out/rq_z.json 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())['rq_3x256']
print(d['fast']['ms'], d['identical'], d['rows_differing'], d['fast']['stats']['records'], d['fast']['stats']['rows_reencoded_exactly'])"; done
DATA=aniso CODEBOOK=trained python tools/bench_rq.py 4000000 gpurun_out/rq_z2.json 2>&1 | tail -2 | cut -c1-300
