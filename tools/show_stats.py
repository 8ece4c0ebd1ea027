"""Print a rocprofv3 kernel_stats.csv compactly: python tools/show_stats.py <dir or csv> [n]"""
import csv, glob, os, sys
src = sys.argv[1]
if os.path.isdir(src):
    src = max(glob.glob(src + "/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = list(csv.DictReader(open(src)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms" % (tot / 1e6))
for r in rows[:n]:
    name = r["Name"].replace("mevi::(anonymous namespace)::", "").replace("void ", "")
    print("%-64s %6s calls %9.3f ms %6.2f%%  avg %9.1f us" % (name[:64], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                           float(r["Percentage"]), float(r["AverageNs"]) / 1e3))
