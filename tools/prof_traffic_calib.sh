#!/bin/bash
# Calibration of FETCH_SIZE on this box: the single-query search streams the 13.58 GB f16 corpus image exactly once
# (ip_filter_h1_small_kernel, LDS-DMA 16 B/lane reads -- the access pattern of the big filter kernel), so
# factor = 13.58 GB / (FETCH_SIZE KiB x 1024 per search).  The guide's gfx950 note says 2; measure it.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/traffic_calib; rm -rf $OUT; mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/tools/probe_dense.py 8841823 1 100 > $OUT/fetch.log 2>&1
grep "^v0" $OUT/fetch.log | tail -1
python3 - $OUT <<'PY'
import csv, glob, os, sys
cc = max(glob.glob(f"{sys.argv[1]}/fetch/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
tot, disp = {}, {}
for r in csv.DictReader(open(cc)):
    k = r["Kernel_Name"].replace("mevi::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
    tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
    disp.setdefault(k, set()).add(r["Dispatch_Id"])
image = 8841823 * 768 * 2
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:8]:
    print(f"{k:42s} dispatches {len(disp[k]):3d}  FETCH_SIZE {v * 1024 / 1e9:8.2f} GB over 3 searches -> per search {v * 1024 / 3 / 1e9:7.2f} GB; image {image / 1e9:.2f} GB; factor {image / (v * 1024 / 3):.3f}")
PY
