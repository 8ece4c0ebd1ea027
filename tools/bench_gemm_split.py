"""Split-precision GEMM (csrc/gemm_split.hip) vs the exact f32-MFMA GEMM: accuracy against f64 and rate.
  python tools/bench_gemm_split.py [M N K]..."""
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mevi_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
shapes = [(76906, 2304, 768), (76906, 768, 768), (76906, 3072, 768), (76906, 768, 3072), (81920, 768, 768),
          (8192, 768, 768), (20480, 2304, 768), (1000, 768, 768), (96, 768, 768), (5, 3072, 768)]
if len(sys.argv) > 3:
    shapes = [tuple(int(v) for v in sys.argv[i:i + 3]) for i in range(1, len(sys.argv) - 2, 3)]
g = torch.Generator(device=dev).manual_seed(0)
for M, N, K in shapes:
    x = torch.randn((M, K), device=dev, generator=g) * torch.exp(torch.randn((M, 1), device=dev, generator=g))
    w = torch.randn((N, K), device=dev, generator=g) * K ** -0.5
    ws = ops.split_rows(w)
    exact = ops.linear(x, w)
    xs = ops.split_rows(x)
    fast = ops.linear(xs, ws)
    rows = slice(0, min(M, 512))
    ref = x[rows].double() @ w.double().T
    den = x[rows].double().abs() @ w.double().abs().T
    e_exact = ((exact[rows].double() - ref).abs() / den).max().item()
    e_fast = ((fast[rows].double() - ref).abs() / den).max().item()
    rms_exact = ((exact[rows].double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    rms_fast = ((fast[rows].double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()

    def timeit(fn, n=10):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n

    t_exact = timeit(lambda: ops.linear(x, w))
    t_fast = timeit(lambda: ops.linear(xs, ws))
    t_split = timeit(lambda: ops.split_rows(x))
    fl = 2.0 * M * N * K
    print(f"{M:6d} x {N:5d} x {K:5d}: exact {t_exact * 1e3:8.3f} ms {fl / t_exact / 1e12:6.1f} TF | split {t_fast * 1e3:8.3f} ms "
          f"{fl / t_fast / 1e12:6.1f} TF (x{t_exact / t_fast:4.2f}) + split_rows {t_split * 1e3:6.3f} ms | max err/sum|a||w| exact "
          f"{e_exact:.2e} split {e_fast:.2e} | rel rms exact {rms_exact:.2e} split {rms_fast:.2e}", flush=True)
    # the same rows alone (skinny kernel) and inside the batch (tile stream): identical bits
    if M * N > 1500000:
        few = ops.linear(ops.split_rows(x[:7].contiguous()), ws)
        assert torch.equal(few, fast[:7]), "skinny kernel and tile stream disagree"
print("skinny == tile stream bits: ok")
