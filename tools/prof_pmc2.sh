#!/bin/bash
# instruction-mix counters of the dense filter kernel (own PMC passes, kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
ND=${1:-4000000}
OUT=$R/gpurun_out/pmc2
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT/a -- python3 $R/tools/probe_dense.py $ND > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/b -- python3 $R/tools/probe_dense.py $ND > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -- python3 $R/tools/probe_dense.py $ND > $OUT/c.log 2>&1
for f in $OUT/*.log; do tail -n 1 $f | cut -c1-200; done
