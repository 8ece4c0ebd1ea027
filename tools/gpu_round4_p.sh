# PMC of attention_mfma16_kernel inside an NCI pass (what bounds it at 3 TB/s?)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/r4p; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/sq1 -- python3 $R/tools/bench_nci.py 6980 6980 4 32 > $OUT/sq1.log 2>&1
echo rc=$?
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --output-format csv -d $OUT/sq2 -- python3 $R/tools/bench_nci.py 6980 6980 4 32 > $OUT/sq2.log 2>&1
echo rc=$?
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/tcc -- python3 $R/tools/bench_nci.py 6980 6980 4 32 > $OUT/tcc.log 2>&1
echo rc=$?
python3 - <<'P'
import csv,glob,os,collections
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for name in ("sq1","sq2","tcc"):
    fs=glob.glob(f"{R}/gpurun_out/r4p/{name}/**/*counter_collection.csv",recursive=True)
    if not fs: print(name,"no csv"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"]
        key="mfma16" if "attention_mfma16" in k else "few_keys" if "few_keys" in k else "rmsnorm_split" if "rmsnorm_split" in k else None
        if key is None: continue
        agg[key][r["Counter_Name"]]+=float(r["Counter_Value"])
    for key,d in agg.items():
        print(name,key," ".join(f"{c}={v:.4g}" for c,v in sorted(d.items())))
P
