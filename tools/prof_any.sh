#!/bin/bash
# rocprofv3 kernel stats of one tools/ script: bash tools/prof_any.sh <name> <script.py> [args...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
OUT=$R/gpurun_out/$NAME
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/"$@" > $OUT/log.txt 2>&1
grep -v "^[EW]2026" $OUT/log.txt | tail -n 3
