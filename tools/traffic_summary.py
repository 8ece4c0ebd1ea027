"""Per-launch HBM-side traffic of the filter kernel from the PMC passes of tools/prof_traffic.sh.
FETCH_SIZE is in KiB and, on gfx950, reports half the bytes of 16 B/lane reads (MI355X_MICROARCH.md, HBM): x2."""
import csv, glob, json, os, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/traffic"
out = {}
for name in ("fetch", "tcc"):
    cc = max(glob.glob(f"{root}/{name}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    tot, n = {}, set()
    for r in csv.DictReader(open(cc)):
        if "ip_filter_h1" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            n.add(r["Dispatch_Id"])
    out[name] = (tot, len(n))
fetch, nl = out["fetch"]
tcc, _ = out["tcc"]
res = {
    "kernel": "ip_filter_h1_kernel", "launches": nl,
    "fetch_size_kib_sum": fetch["FETCH_SIZE"],
    "hbm_side_bytes_per_launch": fetch["FETCH_SIZE"] * 1024 * 2 / nl,
    "correction": "FETCH_SIZE (KiB) x 2: gfx950 tallies the 128-byte requests of 16 B/lane reads at 64 B",
    "l2_hit_rate": tcc["TCC_HIT_sum"] / (tcc["TCC_HIT_sum"] + tcc["TCC_MISS_sum"]),
    "l2_requests_per_launch": (tcc["TCC_HIT_sum"] + tcc["TCC_MISS_sum"]) / nl,
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | TCC_HIT_sum TCC_MISS_sum -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline",
    "note": "memory-side requests of the L2: Infinity Cache hits are counted (the 10.7 MB query image and the re-read corpus tiles live there), so this is an upper bound of the HBM bytes",
}
print(json.dumps(res, indent=1))
