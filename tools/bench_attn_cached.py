#!/usr/bin/env python3
"""A/B of the decode-step self-attention over ancestor-indexed caches (ops.attention_cached; t5.py DecoderStack.step) at the
shapes of the NCI beam search: n = queries x beams rows, 12 heads x 64, tk = 1 .. 5 cached positions.

    python tools/bench_attn_cached.py            # runs itself once per kernel form and compares output hashes + times

MEVI_ATTN_FEW_KEYS=direct is the per-lane row walk (rounds 1-2), the default the LDS-transposed form (round 3)."""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch

    from mevi_amd import ops

    dev = torch.device("cuda", 0)
    nq, R, T = int(os.environ.get("NQ", "6980")), 10, 9
    H, dh = int(os.environ.get("HEADS", "12")), int(os.environ.get("DH", "64"))     # HEADS=8 DH=96: the adaptor's heads
    n = int(os.environ.get("NROWS", nq * R))       # NROWS: a row count that is not a multiple of anything (ragged last wave)
    g = torch.Generator(device="cpu").manual_seed(5)
    cache = torch.randn((n, T, 2 * H * dh), generator=g).to(dev)
    q = torch.randn((n, H * dh), generator=g).to(dev)
    bias = torch.randn((H, T, T), generator=g).to(dev)
    out = {}
    for tk in (1, 2, 3, 4, 5, 6, 7, 8):
        # ancestors: position j of row (query, r) lives in a row of the same query; position tk - 1 is the row itself
        # ANC=beam (default): a beam-search-like genealogy -- position 0 is shared by all beams of a query and the number of
        # distinct ancestors grows with the position; ANC=random: independent ancestors (every row distinct: worst case)
        if os.environ.get("ANC", "beam") == "random":
            anc = torch.randint(0, R, (n, tk), generator=g, dtype=torch.int32)
        else:
            anc = torch.stack([torch.randint(0, min(R, 1 + 3 * jj), (n,), generator=g, dtype=torch.int32) for jj in range(tk)], 1)
        base = (torch.arange(n, dtype=torch.int32) // R * R)[:, None]
        key_rows = (anc + base)
        key_rows[:, tk - 1] = torch.arange(n, dtype=torch.int32)
        key_rows.clamp_(max=n - 1)
        key_rows = key_rows.to(dev).contiguous()
        k, v = cache[:, :, :H * dh], cache[:, :, H * dh:]
        f = lambda: ops.attention_cached(q, k, v, key_rows, H, bias=bias, q_pos0=tk - 1, causal=True, scale=dh ** -0.5 if dh != 64 else 1.0)      # noqa: E731
        o = f()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            f()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / 20 * 1e3
        out[tk] = {"ms": round(ms, 4), "sha": hashlib.sha1(o.cpu().numpy().tobytes()).hexdigest()[:16]}
    print("ATTN_JSON " + json.dumps(out), flush=True)


if __name__ == "__main__":
    if os.environ.get("ATTN_CHILD") == "1":
        child()
        sys.exit(0)
    res = {}
    for mode in ("direct", "lds"):
        env = dict(os.environ, ATTN_CHILD="1", MEVI_ATTN_FEW_KEYS=mode)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("ATTN_JSON ")]
        if not line:
            print(r.stdout[-2000:], r.stderr[-2000:])
            sys.exit(1)
        res[mode] = json.loads(line[0][len("ATTN_JSON "):])
    ok = True
    for tk in res["direct"]:
        d, l = res["direct"][tk], res["lds"][tk]
        ok = ok and d["sha"] == l["sha"]
        print(f"tk={tk}: direct {d['ms']:.3f} ms, lds {l['ms']:.3f} ms, same bits: {d['sha'] == l['sha']}")
    sys.exit(0 if ok else 1)
