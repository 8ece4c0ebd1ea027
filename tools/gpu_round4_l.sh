R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/r4k/tower; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/prof_latency.py tower > $OUT/log.txt 2>&1
python3 - <<'P'
import csv,glob,collections,os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
f=glob.glob(R+'/gpurun_out/r4k/tower/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'skinny' in r['Kernel_Name']]
g=collections.defaultdict(list)
for r in rows:
    g[r['Grid_Size_X']].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(g.items()):
    h=collections.Counter(int(x) for x in v); print(k, len(v), sorted(h.items()))
P
python3 $R/tools/show_stats.py $OUT 8
