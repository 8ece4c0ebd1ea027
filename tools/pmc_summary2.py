"""Per-variant PMC table: largest ip_filter dispatch of each distinct kernel name."""
import csv, glob, sys
from collections import defaultdict
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc2"
for d in sorted(glob.glob(root + "/*/")):
    cc = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not cc: continue
    rows = defaultdict(dict); names = {}; grid = {}
    for r in csv.DictReader(open(cc[0])):
        rows[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"]; grid[r["Dispatch_Id"]] = int(r["Grid_Size"])
    best = {}
    for did, n in names.items():
        if "ip_filter" in n and (n not in best or grid[did] > grid[best[n]]): best[n] = did
    print("==", d)
    ks = sorted(best)
    ctrs = sorted({c for k in ks for c in rows[best[k]]})
    print("%-28s" % "counter", " ".join("%14s" % k.split("ip_filter_kernel")[1][:12] for k in ks))
    for c in ctrs:
        print("%-28s" % c, " ".join("%14.4g" % rows[best[k]].get(c, float("nan")) for k in ks))
