"""sha256 of the outputs of the short-sequence attention paths on seeded inputs (for A/B of two builds of the library:
MEVI_PROBE_LIB=... python tools/probe_attn_bits.py): padded + masked and packed self-attention with a bias table, decode-step
cross-attention over shared packed / padded keys (with an empty group), f32 and split-image outputs."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mevi_amd import hip as _hip  # noqa: E402
if os.environ.get("MEVI_PROBE_LIB"):
    _hip.LIB = os.path.abspath(os.environ["MEVI_PROBE_LIB"])
from mevi_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
H, dh = 12, 64
hd = H * dh
h = hashlib.sha256()


def add(t):
    t = t.img if hasattr(t, "img") else t
    h.update(t.detach().contiguous().cpu().numpy().tobytes())


def rn(*s):
    return torch.randn(s, device=dev, generator=g)


rng = np.random.default_rng(0)
for S in (7, 16, 17, 32):
    nb = 37
    qkv = rn(nb, S, 3 * hd)
    bias = rn(H, S, S)
    lens = rng.integers(1, S + 1, size=nb)
    mask = torch.from_numpy((np.arange(S)[None] < lens[:, None]).astype(np.int64)).to(dev)
    for sb in (None, 3.0):
        add(ops.attention(qkv[:, :, :hd], qkv[:, :, hd:2 * hd], qkv[:, :, 2 * hd:], H, bias=bias, key_mask=mask, split_bound=sb))
    off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)])).to(dev)
    packed = torch.cat([qkv[i, :lens[i]] for i in range(nb)])
    for sb in (None, 3.0):
        add(ops.attention_varlen(packed[:, :hd], packed[:, hd:2 * hd], packed[:, 2 * hd:], off, int(lens.max()), H, bias=bias, split_bound=sb))
for kv_div, longest in ((10, 32), (10, 12), (1, 20), (4, 16)):
    nqr = 23
    lens = rng.integers(0 if kv_div == 10 else 1, longest + 1, size=nqr)
    lens[0] = longest
    off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)])).to(dev)
    kv = rn(int(lens.sum()), 2 * hd)
    q = rn(nqr * kv_div, 1, hd)
    for sb in (None, 2.0):
        add(ops.attention(q, kv[:, :hd], kv[:, hd:], H, kv_div=kv_div, kv_off=off, kv_longest=longest, split_bound=sb))
    kvp = rn(nqr, longest, 2 * hd)
    mask = torch.from_numpy((np.arange(longest)[None] < np.maximum(lens, 1)[:, None]).astype(np.int64)).to(dev)
    add(ops.attention(q, kvp[:, :, :hd], kvp[:, :, hd:], H, kv_div=kv_div, key_mask=mask))
torch.cuda.synchronize()
print("attention bits", h.hexdigest())
