#!/bin/bash
# round 3, seq2seq arm: staged few-keys attention + attention contexts as split images; tests, then A/B of the NCI / tower legs
set -x
python -m pytest tests/test_ops_gpu.py tests/test_t5_gpu.py tests/test_e2e_gpu.py -q -m gpu -x 2>&1 | tail -5
HEADS=8 DH=96 python tools/bench_attn_cached.py 2>&1 | tail -8
for mode in "MEVI_ATTN_FEW_KEYS=direct MEVI_ATTN_CTX=f32" ""; do
  echo "=== $mode"
  env $mode python tools/bench_chain.py 2>&1 | grep -v "^{" | tail -8
done
