#!/bin/bash
# round 4: the 16x16x32 stream probe, then the GPU suite (no -x)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4b
timeout 300 tools/probes/mfma16_probe > gpurun_out/r4b/mfma16_probe.txt 2>&1; echo "probe rc=$?"; cat gpurun_out/r4b/mfma16_probe.txt
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r4b/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r4b/pytest.log
