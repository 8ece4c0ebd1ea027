#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/clk
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/a -- python3 $R/tools/probe_dense.py 4000000 > $OUT/a.log 2>&1
tail -n 4 $OUT/a.log
