"""Per-launch HBM-side traffic of the filter kernel from the PMC passes of tools/prof_traffic.sh.
FETCH_SIZE is in KiB and, on gfx950, reports half the bytes of 16 B/lane reads (MI355X_MICROARCH.md, HBM): x2."""
import csv, glob, hashlib, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILTER_SOURCES = ("mevi_amd/csrc/ip_topk.hip", "mevi_amd/csrc/mfma_pp_f16x16.h", "mevi_amd/csrc/mfma_pp_f16.h")


def filter_source_sha():
    """sha256 over the sources of the dense filter kernel: recorded with a PMC pass and compared by bench.py with the sources
    it runs, so that a traffic figure older than the kernel says so in the bench line."""
    h = hashlib.sha256()
    for f in FILTER_SOURCES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def main(root):
    out = {}
    for name in ("fetch", "tcc"):
        cc = max(glob.glob(f"{root}/{name}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
        tot, n, kernels = {}, set(), set()
        for r in csv.DictReader(open(cc)):
            if "ip_filter_h1" in r["Kernel_Name"] and "small" not in r["Kernel_Name"]:
                tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                n.add(r["Dispatch_Id"])
                kernels.add(re.search(r"ip_filter_h1\w*", r["Kernel_Name"]).group(0))
        out[name] = (tot, len(n), sorted(kernels))
    fetch, nl, kernels = out["fetch"]
    tcc, _, _ = out["tcc"]
    res = {
        "kernel": ", ".join(kernels), "launches": nl,
        "fetch_size_kib_sum": fetch["FETCH_SIZE"],
        "hbm_side_bytes_per_launch": fetch["FETCH_SIZE"] * 1024 * 2 / nl,
        "correction": "FETCH_SIZE (KiB) x 2: gfx950 tallies the 128-byte requests of 16 B/lane reads at 64 B",
        "l2_hit_rate": tcc["TCC_HIT_sum"] / (tcc["TCC_HIT_sum"] + tcc["TCC_MISS_sum"]),
        "l2_requests_per_launch": (tcc["TCC_HIT_sum"] + tcc["TCC_MISS_sum"]) / nl,
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | TCC_HIT_sum TCC_MISS_sum -- python3 bench.py --steps 2 --warmup 1 "
                   "--no-cpu-baseline --no-seq2seq-legs (tools/prof_traffic.sh)",
        "note": "memory-side requests of the L2: Infinity Cache hits are counted (the 10.7 MB query image and the re-read corpus "
                "tiles live there), so this is an upper bound of the HBM bytes",
        "filter_source_sha256": filter_source_sha(),
    }
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/traffic")
