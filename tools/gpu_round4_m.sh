# (3,256) seq2seq arm: where the time goes, with the default prefix-table budget and with 64 GiB (position 2's head matrices tabled)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for G in 6 64; do
  OUT=$R/gpurun_out/r4m/nci_3x256_$G; rm -rf $OUT; mkdir -p $OUT
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/bench_nci.py 6980 6980 3 256 $G > $OUT/log.txt 2>&1
  tail -n 2 $OUT/log.txt
  python3 $R/tools/show_stats.py $OUT 16
done
