#!/bin/bash
# SQ counters per filter launch (chunk) of one search
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/r4e; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-seq2seq-legs > $OUT/sq.log 2>&1
tail -c 300 $OUT/sq.log
python3 $R/tools/pmc_filter_chunks.py $OUT/sq
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INSTS_VALU_MFMA_F16 --output-format csv -d $OUT/sq2 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-seq2seq-legs > $OUT/sq2.log 2>&1
tail -c 200 $OUT/sq2.log
python3 $R/tools/pmc_filter_chunks.py $OUT/sq2
