#!/usr/bin/env python3
"""f32 GEMM (mevi_gemm_nt_f32) at the shapes of the T5 stacks: TFLOP/s per shape."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mevi_amd import hip, ops  # noqa: E402

if os.environ.get("MEVI_PROBE_LIB"):
    hip.LIB = os.path.abspath(os.environ["MEVI_PROBE_LIB"])
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for M in [int(x) for x in os.environ.get("GEMM_M", "512,5120,16384,65536").split(",")]:
    for K, N in ((768, 768), (768, 2304), (768, 3072), (3072, 768), (768, 1536)):
        a = torch.randn((M, K), device=dev, generator=g)
        w = torch.randn((N, K), device=dev, generator=g) * K ** -0.5
        ops.linear(a, w)
        torch.cuda.synchronize()
        reps = max(2, int(2e12 / (2.0 * M * K * N)))
        t = time.perf_counter()
        for _ in range(reps):
            ops.linear(a, w)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        print(f"M={M:6d} K={K:5d} N={N:5d}: {dt*1e3:8.3f} ms  {2.0*M*K*N/dt/1e12:6.1f} TFLOP/s  tiles={((M+255)//256)*((N+127)//128)}", flush=True)
