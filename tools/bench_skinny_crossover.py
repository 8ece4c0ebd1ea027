"""Where the split GEMM's latency kernel (one workgroup per 32 x 32 outputs) hands over to the tile stream: time per call
of both at the seq2seq arm's projection shapes.  Run once per kernel:
  MEVI_GEMM_SKINNY_MAX=0 python tools/bench_skinny_crossover.py ; MEVI_GEMM_SKINNY_MAX=1000000000 python tools/bench_skinny_crossover.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mevi_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
tag = "tile stream" if os.environ.get("MEVI_GEMM_SKINNY_MAX") == "0" else "latency kernel"
for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072)):
    w = ops.split_rows(torch.randn((N, K), device=dev, generator=g) * K ** -0.5)
    for M in (10, 64, 128, 256, 512, 640, 1024, 2048, 4096):
        x = ops.split_rows(torch.randn((M, K), device=dev, generator=g))
        for _ in range(3):
            ops.linear(x, w)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100):
            ops.linear(x, w)
        b.record()
        torch.cuda.synchronize()
        print(f"{tag}: {M:5d} x {N:5d} x {K:5d}  {a.elapsed_time(b) * 10:8.1f} us", flush=True)
