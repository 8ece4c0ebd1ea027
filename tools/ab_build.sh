#!/bin/bash
# Build the HIP library of a git ref (or WORK for the working tree) as tools/probes/ab/lib<name>.so, for
# same-device A/B timing:  MEVI_PROBE_LIB=tools/probes/ab/lib<name>.so python tools/probe_dense.py ...
set -e
REF=$1; NAME=$2
R=$(cd "$(dirname "$0")/.." && pwd)
T=/tmp/ab/$NAME; rm -rf $T; mkdir -p $T $R/tools/probes/ab
if [ "$REF" = WORK ]; then mkdir -p $T/mevi_amd; cp -r $R/mevi_amd/csrc $T/mevi_amd/; cp -r $R/include $T/;
else (cd $R && git archive $REF mevi_amd/csrc include) | tar -x -C $T; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function $T/mevi_amd/csrc/*.hip -o $R/tools/probes/ab/lib$NAME.so
echo built $R/tools/probes/ab/lib$NAME.so
