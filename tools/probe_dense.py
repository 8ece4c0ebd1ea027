"""Quick timing probe of the dense arm (not the bench): python tools/probe_dense.py [nd] [nq] [k]."""
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mevi_amd import dense, hip  # noqa: E402
INDEXED = os.environ.get("INDEXED", "1") == "1"
MEAN = float(os.environ.get("MEAN", "0.02"))  # common component of every row (0.2: cosines ~0.94, dense-retriever like)
if os.environ.get("MEVI_PROBE_LIB"):  # A/B timing of another build of the library on the same device
    hip.LIB = os.path.abspath(os.environ["MEVI_PROBE_LIB"])

nd = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 6980
k = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
dim = 768
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
q = torch.randn((nq, dim), device=dev, generator=g)
if os.environ.get("MEAN"):
    q = 0.05 * q + MEAN
d = torch.empty((nd, dim), device=dev)
for a in range(0, nd, 1 << 20):
    d[a:a + (1 << 20)] = 0.05 * torch.randn((min(1 << 20, nd - a), dim), device=dev, generator=g) + MEAN
hip.lib().mevi_ip_topk_set_profiling(1)
index = dense.DenseIndex(d) if INDEXED else None
for it in range(3):
    v = 0
    torch.cuda.synchronize()
    t = time.time()
    s, i = (index.search(q, k) if INDEXED else dense.ip_topk(q, d, k))
    torch.cuda.synchronize()
    dt = time.time() - t
    st = hip.IpTopkStats()
    hip.lib().mevi_ip_topk_get_stats(st)
    print(f"v{v} nd={nd} nq={nq} k={k}: {dt*1e3:.1f} ms  {2*nq*nd*dim/dt/1e12:.1f} TFLOP/s  "
          f"chunks={st.n_chunks} failed={st.n_failed_queries} filter={st.filter_ms:.1f}ms "
          f"({st.filter_flops/st.filter_ms/1e9:.1f} TF) compact={st.compact_ms:.1f}ms", flush=True)
