# fused logits kernel + indexed adaptor cache: NCI timing at both code shapes, A/B on one box by MEVI_HEAD_LOGITS / MEVI_ADAPTOR_CACHE
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in 1 2; do
python tools/bench_nci.py 6980 6980 4 32 | tail -1
MEVI_HEAD_LOGITS=columns python tools/bench_nci.py 6980 6980 4 32 | tail -1
MEVI_HEAD_LOGITS=columns MEVI_ADAPTOR_CACHE=copy python tools/bench_nci.py 6980 6980 4 32 | tail -1
done
python tools/bench_nci.py 6980 6980 3 256 | tail -1
MEVI_HEAD_LOGITS=columns python tools/bench_nci.py 6980 6980 3 256 | tail -1
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/r4n/nci; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/bench_nci.py 6980 6980 4 32 > $OUT/log.txt 2>&1
python3 $R/tools/show_stats.py $OUT 24
