"""sha256 of split_rows / gather_rows / scatter_rows outputs on seeded inputs (A/B of two builds: MEVI_PROBE_LIB=...)."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mevi_amd import hip as _hip  # noqa: E402
if os.environ.get("MEVI_PROBE_LIB"):
    _hip.LIB = os.path.abspath(os.environ["MEVI_PROBE_LIB"])
from mevi_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
h = hashlib.sha256()
for rows, dim in ((1000, 768), (37, 768), (513, 1024), (200, 64), (90, 2048), (50, 3072)):
    x = torch.randn((rows, dim + 8), device=dev, generator=g)[:, 4:4 + dim] * torch.exp(torch.randn((rows, 1), device=dev, generator=g) * 3)
    if dim % 8 == 0 and dim >= 64:
        sr = ops.split_rows(x if dim != 768 else x.contiguous())
        for t in (sr.img, sr.exp, sr.norm):
            h.update(t.contiguous().cpu().numpy().tobytes())
        sr = ops.split_rows(x)
        for t in (sr.img, sr.exp, sr.norm):
            h.update(t.contiguous().cpu().numpy().tobytes())
    idx = torch.randint(0, rows, (rows + 11,), device=dev, generator=g)
    xc = x.contiguous()
    got = ops.gather_rows(xc, idx)
    h.update(got.cpu().numpy().tobytes())
    assert torch.equal(got, xc[idx])
    perm = torch.randperm(rows, device=dev, generator=g)
    out = torch.zeros((rows, dim), device=dev)
    ops.scatter_rows(xc, perm, out)
    ref = torch.zeros_like(out)
    ref[perm] = xc
    assert torch.equal(out, ref)
    h.update(out.cpu().numpy().tobytes())
torch.cuda.synchronize()
print("rows bits", h.hexdigest())
