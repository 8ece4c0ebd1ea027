"""Diagnostic: per-phase s_memtime stamps of one ping-pong workgroup (VARIANT 3 build)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mevi_amd import dense, hip
L = hip.lib()
L.mevi_debug_set_variant.argtypes = [ctypes.c_int]
L.mevi_debug_read_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
q = torch.randn((6980, 768), device=dev)
d = torch.randn((1_000_000, 768), device=dev) * 0.05
dense.ip_topk(q, d, 1000)
L.mevi_debug_set_variant(3)
dense.ip_topk(q, d, 1000)
torch.cuda.synchronize()
st = np.zeros(512, np.uint64)
L.mevi_debug_read_stamps(st.ctypes.data_as(ctypes.c_void_p))
st = st.reshape(8, 8, 8).astype(np.int64)
t0 = st[:, :, :6][st[:, :, :6] > 0].min()
for w in (0, 1, 4, 5):
    print("wave", w, "(G%d)" % (w // 4))
    for s in range(1, 6):
        r = st[w, s, :4] - t0
        nxt = st[w, s + 1, 0] - t0
        print("   s=%d" % s, " ".join("%7d" % x for x in r), "| deltas", " ".join("%6d" % x for x in np.diff(np.append(r, nxt))))
