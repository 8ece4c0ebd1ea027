"""Per-launch durations of the dense filter kernel from a rocprofv3 --kernel-trace CSV: the chunk schedule of one search
(8 launches of growing size) with the loop-only time each chunk's FLOP would take at a given tile-loop rate.
usage: python tools/filter_launches.py <dir with *kernel_trace.csv> [loop TFLOP/s] [steps to skip]"""
import csv, glob, os, sys

d = sys.argv[1]
rate = float(sys.argv[2]) if len(sys.argv) > 2 else 1316.0
src = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(src)))
f = [r for r in rows if "ip_filter_h1" in r["Kernel_Name"] and "small" not in r["Kernel_Name"]]
f.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in f]
grid = [int(r["Grid_Size_X"] if "Grid_Size_X" in r else r["Grid_Size"]) for r in f]
name = f[0]["Kernel_Name"][:60] if f else "?"
per = 8
n = len(dur) // per
print(name, len(dur), "launches =", n, "searches")
# chunk rows are not in the trace: print the per-position mean over the searches (warm-up search first)
for i in range(per):
    xs = [dur[s * per + i] for s in range(1, n)] or [dur[i]]
    print("launch %d: %.3f ms (mean of %d), grid %d" % (i, sum(xs) / len(xs), len(xs), grid[i]))
tot = [sum(dur[s * per:(s + 1) * per]) for s in range(1, n)] or [sum(dur[:per])]
print("filter total per search: %.3f ms; loop-only at %.0f TFLOP/s: %.3f ms" % (sum(tot) / len(tot), rate, 2.0 * 6980 * 8841823 * 768 / rate / 1e9))
