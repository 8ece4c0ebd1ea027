#!/bin/bash
# PMC passes for the dense filter kernel (run on the GPU box via gpurun).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
ND=${1:-2000000}
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/sq -- python3 $R/tools/probe_dense.py $ND > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/tcc -- python3 $R/tools/probe_dense.py $ND > $OUT/tcc.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/tools/probe_dense.py $ND > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/lds -- python3 $R/tools/probe_dense.py $ND > $OUT/lds.log 2>&1
find $OUT -name "*.csv" | head -20
for f in $OUT/*.log; do tail -n 2 $f; done
