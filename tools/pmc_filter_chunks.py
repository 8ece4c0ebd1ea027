"""Per-launch SQ counters of the dense filter (one search = 8 launches): instruction counts per wave-tile by chunk, from a
rocprofv3 --pmc counter_collection CSV.  usage: python tools/pmc_filter_chunks.py <dir>"""
import csv, glob, sys
from collections import defaultdict, OrderedDict

d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
rows = OrderedDict()
for r in csv.DictReader(open(cc)):
    if "ip_filter_h1" not in r["Kernel_Name"] or "small" in r["Kernel_Name"]:
        continue
    rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)[-8:]
names = sorted({k for i in ids for k in rows[i]})
print("launch " + " ".join("%14s" % n[-14:] for n in names))
for j, i in enumerate(ids):
    print("%6d " % j + " ".join("%14.4g" % rows[i].get(n, float("nan")) for n in names))
