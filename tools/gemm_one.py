"""A few split-GEMM shapes back to back for a kernel trace (per-kernel GPU time instead of a host loop's):
python tools/gemm_one.py MxNxK [MxNxK ...]   (each shape: 20 calls with a residual operand)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mevi_amd import hip as _hip  # noqa: E402
if os.environ.get("MEVI_PROBE_LIB"):
    _hip.LIB = os.path.abspath(os.environ["MEVI_PROBE_LIB"])
from mevi_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for spec in sys.argv[1:]:
    M, N, K = (int(v) for v in spec.split("x"))
    g = torch.Generator(device=dev).manual_seed(N + K)
    w = ops.weight_split(torch.randn((N, K), device=dev, generator=g) * K ** -0.5)
    x = ops.split_rows(torch.randn((M, K), device=dev, generator=g))
    r = torch.randn((M, N), device=dev, generator=g)
    for _ in range(3):
        ops.linear(x, w, residual=r)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        ops.linear(x, w, residual=r)
    b.record()
    torch.cuda.synchronize()
    print("%s: %.1f us per call (host loop)" % (spec, a.elapsed_time(b) / 20 * 1e3), flush=True)
