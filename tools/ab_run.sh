#!/bin/bash
# same-device A/B: tools/ab_run.sh <nd> <name> <name> ...   (libraries built by tools/ab_build.sh)
ND=$1; shift
for i in 1 2; do for v in "$@"; do echo "== $v"; MEVI_PROBE_LIB=tools/probes/ab/lib$v.so timeout 300 python tools/probe_dense.py $ND 2>&1 | tail -1; done; done
