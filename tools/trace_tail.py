"""Per-kernel summary of the LAST phase of a rocprofv3 --kernel-trace run: the kernels after the last idle gap of at least
`gap_ms` (default 200) -- tools/bench_nci.py / bench_tower.py sleep before their timed pass when TRACE_GAP=1, so the table holds the
timed pass alone (no warm-up pass, no prefix-table build).   python tools/trace_tail.py <dir or kernel_trace.csv> [out.csv] [gap_ms]"""
import csv
import glob
import os
import sys

src = sys.argv[1]
if os.path.isdir(src):
    src = max(glob.glob(src + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
gap = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 200e6
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(src))))
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - rows[i - 1][1] >= gap:
        cut = i
rows = rows[cut:]
if os.environ.get("TIMELINE"):       # every launch of the phase: start offset, duration, idle gap before it
    t0, prev = rows[0][0], rows[0][0]
    for s_, e_, n_ in rows:
        print("%10.1f us  +%9.1f us  (gap %7.1f)  %s" % ((s_ - t0) / 1e3, (e_ - s_) / 1e3, (s_ - prev) / 1e3,
                                                     n_.replace("mevi::(anonymous namespace)::", "")[:90]))
        prev = e_
agg = {}
for s, e, n in rows:
    n = n.replace("mevi::(anonymous namespace)::", "").replace("void ", "")
    a = agg.setdefault(n, [0, 0])
    a[0] += 1
    a[1] += e - s
tot = sum(a[1] for a in agg.values())
span = rows[-1][1] - rows[0][0] if rows else 0
print("last phase: %d kernels, kernel time %.3f ms, span %.3f ms" % (len(rows), tot / 1e6, span / 1e6))
table = sorted(agg.items(), key=lambda kv: -kv[1][1])
for n, (c, t) in table[:int(os.environ.get("TOP", "16"))]:
    print("%-72s %5d calls %9.3f ms %6.2f%%  avg %8.1f us" % (n[:72], c, t / 1e6, 100.0 * t / max(tot, 1), t / c / 1e3))
if len(sys.argv) > 2 and sys.argv[2] != "-":
    with open(sys.argv[2], "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for n, (c, t) in table:
            w.writerow([n[:160], c, t, round(t / c, 1), round(100.0 * t / max(tot, 1), 3)])
