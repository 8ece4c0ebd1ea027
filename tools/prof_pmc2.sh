#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc2
rm -rf $OUT; mkdir -p $OUT
export VARIANTS=${VARIANTS:-4,7}
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_ANY --output-format csv -d $OUT/lds -- python3 $R/tools/probe_dense.py 1000000 > $OUT/lds.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/sq -- python3 $R/tools/probe_dense.py 1000000 > $OUT/sq.log 2>&1
tail -n 3 $OUT/lds.log
