import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from mevi_amd import nci, t5, ops
from oracle import t5 as ot5
import torch.nn.functional as F
g = np.load("tests/golden/g2_t5_tower.npz")
cfg = json.loads(str(g["cfg"]))
W = nci.load_npz_weights(g)
cuda = torch.device("cuda:0")
tower = t5.TwinTower(W, device=cuda, **cfg)
ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
d = tower.d
B, S = ids.shape
enc = tower.encoder
x = ops.gather_rows(tower.shared, ids.to(cuda).reshape(-1))
xo = W["shared.weight"][ids]
print("emb", (x.cpu().view(B,S,-1) - xo).abs().max().item())
L = enc.layers[0]
h = ops.rmsnorm(x, L["ln0"], d.eps)
ho = ot5.rmsnorm(xo, W["encoder.block.0.layer.0.layer_norm.weight"], d.eps)
print("rms", (h.cpu().view(B,S,-1) - ho).abs().max().item())
qkv = ops.linear(h, L["wqkv"]).view(B, S, 3 * d.inner)
p = "encoder.block.0.layer.0.SelfAttention"
qo = ho @ W[p+".q.weight"].T; ko = ho @ W[p+".k.weight"].T; vo = ho @ W[p+".v.weight"].T
print("q", (qkv[:,:,:d.inner].cpu() - qo).abs().max().item(), "k", (qkv[:,:,d.inner:2*d.inner].cpu() - ko).abs().max().item(), "v", (qkv[:,:,2*d.inner:].cpu()-vo).abs().max().item())
bias = enc.bias(S)
bo = ot5.position_bias(W, p, S, S, True)
print("bias", (bias.cpu() - bo[0]).abs().max().item())
ctx = ops.attention(qkv[:, :, :d.inner], qkv[:, :, d.inner:2 * d.inner], qkv[:, :, 2 * d.inner:], d.num_heads, bias=bias, key_mask=mask.to(cuda))
H = d.num_heads
sc = ot5._heads(qo, H) @ ot5._heads(ko, H).transpose(-1,-2) + bo + (1.0 - mask[:, None, None, :].float()) * -1e9
co = (F.softmax(sc, -1) @ ot5._heads(vo, H)).transpose(1,2).reshape(B,S,-1)
print("ctx", (ctx.cpu() - co).abs().max().item())
q_, k_, v_ = qkv[:, :, :d.inner].contiguous(), qkv[:, :, d.inner:2 * d.inner].contiguous(), qkv[:, :, 2 * d.inner:].contiguous()
ctx2 = ops.attention(q_, k_, v_, d.num_heads, bias=bias, key_mask=mask.to(cuda))
print("ctx contiguous", (ctx2.cpu() - co).abs().max().item())
ctx3 = ops.attention(q_, k_, v_, d.num_heads, bias=None, key_mask=None)
sc3 = ot5._heads(qo, H) @ ot5._heads(ko, H).transpose(-1,-2)
co3 = (F.softmax(sc3, -1) @ ot5._heads(vo, H)).transpose(1,2).reshape(B,S,-1)
print("ctx nobias nomask", (ctx3.cpu() - co3).abs().max().item())
ctx4 = ops.attention(q_, k_, v_, d.num_heads, bias=bias, key_mask=None)
co4 = (F.softmax(sc3 + bo, -1) @ ot5._heads(vo, H)).transpose(1,2).reshape(B,S,-1)
print("ctx bias nomask", (ctx4.cpu() - co4).abs().max().item())
print("per batch err", (ctx2.cpu() - co).abs().amax((1,2)))
print(mask.sum(1))
