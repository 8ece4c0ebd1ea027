#!/bin/bash
# clock and matrix-pipe utilisation of the f32 GEMM at the seq2seq arm's shapes (PMC pass, kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/clk_gemm
rm -rf $OUT; mkdir -p $OUT
export GEMM_M=76906
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/a -- python3 $R/tools/bench_gemm.py > $OUT/a.log 2>&1
grep "TFLOP" $OUT/a.log | tail -5
