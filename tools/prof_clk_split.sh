#!/bin/bash
# clock and matrix-pipe utilisation of the split-precision GEMM at the seq2seq arm's shapes (PMC pass, kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/clk_split
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/a -- python3 $R/tools/bench_gemm_split.py 76906 2304 768 76906 768 3072 > $OUT/a.log 2>&1
grep " x " $OUT/a.log | cut -c1-110
