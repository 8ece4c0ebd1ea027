import csv, glob, sys
from collections import defaultdict
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/clk"
match = sys.argv[2] if len(sys.argv) > 2 else "ip_filter"   # kernel-name substring
import os
cc = max(glob.glob(root + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
kt = max(glob.glob(root + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
rows = defaultdict(dict); names = {}; grid = {}
for r in csv.DictReader(open(cc)):
    rows[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"]; grid[r["Dispatch_Id"]] = int(r["Grid_Size"])
best = {}
for did, n in names.items():
    if match in n and (n not in best or dur[did] > dur[best[n]]): best[n] = did  # longest dispatch (persistent kernels share one grid size)
for n, did in sorted(best.items()):
    c = rows[did]; d = dur[did]
    clk = c["GRBM_GUI_ACTIVE"] / 8 / d  # GHz
    util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8)
    print("%-28s dur %.2f ms  clock %.3f GHz  mfma pipe util %.1f%%" % (n[n.index(match):][:28], d / 1e6, clk, 100 * util))
