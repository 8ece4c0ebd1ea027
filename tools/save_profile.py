"""Copy a rocprofv3 --stats kernel summary from gpurun_out/ into profiles/ (names trimmed)."""
import csv, glob, sys
import os
src = max(glob.glob((sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/stats") + "/**/*kernel_stats.csv", recursive=True),
          key=os.path.getmtime)
dst = sys.argv[2]
rows = list(csv.reader(open(src)))
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    for r in rows:
        r[0] = r[0][:140]
        w.writerow(r)
print(dst, len(rows) - 1, "kernels")
