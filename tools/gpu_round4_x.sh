# RQ encode (3,256): kernel times with / without XDIR, and SQ counters of rq_fast_kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  OUT=$R/gpurun_out/r4x/st$v; rm -rf $OUT; mkdir -p $OUT
  MEVI_RQ_XDIRECT=$v rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/bench_rq.py 8841823 $OUT/out.json > $OUT/log.txt 2>&1
  echo "XDIRECT=$v"; python3 $R/tools/show_stats.py $OUT 6 | grep -i "rq_fast\|fixup\|rq_level\|total"
  find $OUT -name "*kernel_trace.csv" -delete
done
OUT=$R/gpurun_out/r4x/pmc; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/a -- python3 $R/tools/bench_rq.py 8841823 $OUT/out.json > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU --output-format csv -d $OUT/b -- python3 $R/tools/bench_rq.py 8841823 $OUT/out.json > $OUT/b.log 2>&1
python3 - <<'P'
import csv,glob,os,collections
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for name in ("a","b"):
    fs=glob.glob(f"{R}/gpurun_out/r4x/pmc/{name}/**/*counter_collection.csv",recursive=True)
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"]
        if "rq_fast_kernel" not in k: continue
        key="K256" if "ILi8ELi8" in k or "<8, 8" in k else "K32"
        agg[key][r["Counter_Name"]]+=float(r["Counter_Value"])
    for key,d in agg.items(): print(name,key," ".join(f"{c}={v:.4g}" for c,v in sorted(d.items())))
P
rm -rf $R/gpurun_out/r4x/pmc/a $R/gpurun_out/r4x/pmc/b
