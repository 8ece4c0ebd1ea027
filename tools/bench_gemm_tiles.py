"""Split GEMM tile stream by tile height (round 5: 256 / 128 / 64 activation rows x 256 W rows): time per call and a hash of the
output at the seq2seq arm's projection shapes, one child process per setting of MEVI_GEMM_TILE_ROWS (the library reads its
switches once).  The hashes of one shape must agree across settings (a row keeps its bits whatever tile it travels in).

    python tools/bench_gemm_tiles.py            # parent: spawns the children, prints the table
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = ((768, 768), (2304, 768), (3072, 768), (768, 3072))
MS = (873, 1418, 2048, 4096, 8730, 9600, 16384, 34900, 69800, 76906)
MS_SMALL = (16, 64, 128, 256, 512, 873, 1418, 2048, 4096)      # `small`: latency kernel vs tile stream (the crossover in outputs)
if os.environ.get("TILES_SMALL"):
    MS = MS_SMALL


def child():
    import torch

    sys.path.insert(0, ROOT)
    from mevi_amd import hip as _hip
    if os.environ.get("MEVI_PROBE_LIB"):
        _hip.LIB = os.path.abspath(os.environ["MEVI_PROBE_LIB"])
    from mevi_amd import ops

    dev = torch.device("cuda:0")
    out = {}
    for N, K in SHAPES:
        g = torch.Generator(device=dev).manual_seed(N + K)
        w = ops.weight_split(torch.randn((N, K), device=dev, generator=g) * K ** -0.5)
        xall = torch.randn((max(MS), K), device=dev, generator=g)
        res = torch.randn((max(MS), N), device=dev, generator=g)
        for M in MS:
            x = ops.split_rows(xall[:M])
            r = res[:M]
            y = ops.linear(x, w, residual=r)
            h = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:12]
            hs = ops.linear(x, w, relu=True, for_gemm=True)
            h2 = hashlib.sha256(hs.img.cpu().numpy().tobytes() + hs.exp.cpu().numpy().tobytes()).hexdigest()[:12]
            for _ in range(3):
                ops.linear(x, w, residual=r)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 50 if M <= 16384 else 20
            a.record()
            for _ in range(reps):
                ops.linear(x, w, residual=r)
            b.record()
            torch.cuda.synchronize()
            out[f"{M}x{N}x{K}"] = {"us": a.elapsed_time(b) / reps * 1e3, "hash": h, "hash_img": h2}
    print(json.dumps(out))


def main_small():
    res = {}
    for tag, mx in (("latency", "1000000000"), ("stream", "0")):
        env = dict(os.environ, MEVI_GEMM_SKINNY_MAX=mx, TILES_SMALL="1")
        env.pop("MEVI_GEMM_TILE_ROWS", None)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
        if p.returncode != 0:
            print(tag, "FAILED", p.stderr[-2000:])
            return
        res[tag] = json.loads(p.stdout.strip().splitlines()[-1])
    print("%-22s %12s %12s   outputs   bits" % ("M x N x K", "latency us", "stream us"))
    for N, K in SHAPES:
        for M in MS_SMALL:
            k = f"{M}x{N}x{K}"
            same = (res["latency"][k]["hash"], res["latency"][k]["hash_img"]) == (res["stream"][k]["hash"], res["stream"][k]["hash_img"])
            print("%-22s %12.1f %12.1f   %8d   %s" % (k, res["latency"][k]["us"], res["stream"][k]["us"], M * N, "same" if same else "DIFFERENT"),
                  flush=True)


def main():
    res = {}
    for rows in ("256", "128", "64", "auto"):
        env = dict(os.environ)
        env.pop("MEVI_GEMM_TILE_ROWS", None)
        if rows != "auto":
            env["MEVI_GEMM_TILE_ROWS"] = rows
        env["MEVI_GEMM_SKINNY_MAX"] = os.environ.get("MEVI_GEMM_SKINNY_MAX", "0")       # tile stream for every shape
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
        if p.returncode != 0:
            print(rows, "FAILED", p.stderr[-2000:])
            continue
        res[rows] = json.loads(p.stdout.strip().splitlines()[-1])
    print("%-22s %9s %9s %9s %9s   r128  r64   TFLOP/s(auto)  bits" % ("M x N x K", "256", "128", "64", "auto"))
    for N, K in SHAPES:
        for M in MS:
            k = f"{M}x{N}x{K}"
            us = [res[r][k]["us"] if r in res else float("nan") for r in ("256", "128", "64", "auto")]
            same = len({(res[r][k]["hash"], res[r][k]["hash_img"]) for r in res}) == 1
            print("%-22s %9.1f %9.1f %9.1f %9.1f   %.2f  %.2f   %7.1f   %s" % (
                k, *us, us[1] / us[0], us[2] / us[0], 2.0 * M * N * K / us[3] / 1e6, "same" if same else "DIFFERENT"), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    elif len(sys.argv) > 1 and sys.argv[1] == "small":
        main_small()
    else:
        main()
