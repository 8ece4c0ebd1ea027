"""Summarise rocprofv3 counter_collection + kernel_trace CSVs: per kernel, largest dispatch."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
for d in sorted(glob.glob(root + "/*/")):
    cc = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    kt = glob.glob(d + "**/*kernel_trace.csv", recursive=True)
    if not cc:
        continue
    dur = {}
    if kt:
        for r in csv.DictReader(open(kt[0])):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    rows = defaultdict(dict)
    names = {}
    for r in csv.DictReader(open(cc[0])):
        rows[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"]
        rows[r["Dispatch_Id"]]["_grid"] = r.get("Grid_Size", "")
        rows[r["Dispatch_Id"]]["_vgpr"] = r.get("VGPR_Count", "")
        rows[r["Dispatch_Id"]]["_lds"] = r.get("LDS_Block_Size", "")
    # biggest filter dispatch
    best = None
    for did, c in rows.items():
        if "ip_filter" in names[did]:
            g = int(c["_grid"] or 0)
            if best is None or g > best[0]:
                best = (g, did)
    print("==", d)
    if best:
        did = best[1]
        c = rows[did]
        print("kernel", names[did][:60], "grid", c["_grid"], "vgpr", c["_vgpr"], "lds", c["_lds"],
              "dur_us", dur.get(did, (0,))[0] / 1e3)
        for k, v in sorted(c.items()):
            if not k.startswith("_"):
                print(f"   {k:32s} {v:.6g}")
