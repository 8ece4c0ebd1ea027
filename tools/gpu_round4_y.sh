# round 4, end: smoke(), the default bench command (wall time) and --gpus 2 (needs two devices: fails loudly on a 1-GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
T0=$(date +%s.%N); python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2; T1=$(date +%s.%N); echo "smoke wall $(python3 -c "print(round($T1 - $T0, 1))") s"
T0=$(date +%s.%N); python bench.py > gpurun_out/y_bench.json 2> gpurun_out/y_bench.err; T1=$(date +%s.%N); echo "default bench wall $(python3 -c "print(round($T1 - $T0, 1))") s"; python - <<'P'
import json
d=json.loads(open("gpurun_out/y_bench.json").read().strip().splitlines()[-1])
print(len(open("gpurun_out/y_bench.json").read()), d["value"], d["steps"], d["warmup"], d["config"]["chain_c4"]["queries_per_s"], d["index_build"]["rq_encode_3x256"]["ms"], d["index_build"]["passage_tower"]["passages_per_s"])
P
T0=$(date +%s.%N); python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/y_bench2.json 2> gpurun_out/y_bench2.err; T1=$(date +%s.%N); echo "--gpus 2 wall $(python3 -c "print(round($T1 - $T0, 1))") s"; python - <<'P'
import json
d=json.loads(open("gpurun_out/y_bench2.json").read().strip().splitlines()[-1])
print(d["n_gpus"], d["value"], d["scaling"], list(d.get("multi_gpu",{}).keys())[:6], "chain_c5" in d or "chain_c5" in d.get("config",{}))
P
