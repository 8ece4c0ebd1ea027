"""Passage self-attention: f32-MFMA kernel vs split-f16 kernel against a float64 reference, and their times.
  MEVI_ATTN_PASSAGE=f32|h16 is read once per process: run twice."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mevi_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(11)
H, dh = 12, 64
hd = H * dh
for scale_in in (1.0, 3.0):
    nb, S = 4, 100
    q, k, v = (torch.randn((nb, S, hd), device=dev, generator=g) * scale_in for _ in range(3))
    bias = torch.randn((H, S, S), device=dev, generator=g)
    qd, kd, vd = (x.double().view(nb, S, H, dh).transpose(1, 2) for x in (q, k, v))
    s = qd @ kd.transpose(-1, -2) + bias.double()[None]
    ref = (torch.softmax(s, -1) @ vd).transpose(1, 2).reshape(nb * S, hd)
    vmax = float(v.abs().max())
    sr = ops.attention(q, k, v, H, bias=bias, split_bound=vmax)
    got = (sr.img[:, :hd].view(torch.float16).float() + sr.img[:, sr.img.shape[1] // 2:sr.img.shape[1] // 2 + hd].view(torch.float16).float()) * 2.0 ** (-int(sr.exp[0]))
    f32 = ops.attention(q, k, v, H, bias=bias).reshape(nb * S, hd)
    print("input scale %.0f: |s|max %.1f  image-kernel err vs f64 %.3e   f32-out kernel err vs f64 %.3e   vmax %.2f" % (
        scale_in, float(s.abs().max()), float((got.double() - ref).abs().max()), float((f32.double() - ref).abs().max()), vmax))
nb, S = 2048, 128
q, k, v = (torch.randn((nb, S, hd), device=dev, generator=g) for _ in range(3))
bias = torch.randn((H, S, S), device=dev, generator=g)
for name, kw in (("image out", dict(split_bound=float(v.abs().max()))), ("f32 out", {})):
    ops.attention(q, k, v, H, bias=bias, **kw)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        ops.attention(q, k, v, H, bias=bias, **kw)
    torch.cuda.synchronize()
    print("%s: %.3f ms per call (2048 x 128 tokens x 12 heads)" % (name, (time.perf_counter() - t) / 5 * 1e3))
