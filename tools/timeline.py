#!/usr/bin/env python3
"""Print one vector step's kernel timeline (start, end, duration in us; queue) from a rocprofv3 kernel trace
(results .db or *_kernel_trace.csv; the newest one under the given directory), and the per-kernel averages.
Usage: timeline.py <dir-or-file> [step-index]"""
import collections, csv, glob, os, sqlite3, sys
path = sys.argv[1]
if os.path.isdir(path):
    c = sorted(glob.glob(os.path.join(path, "**", "*_kernel_trace.csv"), recursive=True) +
               glob.glob(os.path.join(path, "**", "*.db"), recursive=True), key=os.path.getmtime)
    path = c[-1]
if path.endswith(".db"):
    rows = list(sqlite3.connect(path).execute("select name, start, end, queue_id from kernels order by start"))
else:
    rows = sorted(((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"])
                   for r in csv.DictReader(open(path))), key=lambda r: r[1])
idx = [i for i, r in enumerate(rows) if "k_step" in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
i0, i1 = idx[k], idx[k + 1]
t0 = rows[i0][1]
print(f"{path}\nstep {k}: {(rows[i1][1] - t0) / 1000:.1f} us from k_step to k_step")
for r in rows[i0:i1 + 1]:
    print(f"{(r[1] - t0) / 1000:8.1f} {(r[2] - t0) / 1000:8.1f}  {(r[2] - r[1]) / 1000:6.1f}  q{r[3]}  {r[0][:90]}")
print()
agg = collections.defaultdict(list)
for r in rows:
    agg[r[0]].append(r[2] - r[1])
for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"{len(v):6d} x {sum(v) / len(v) / 1000:8.2f} us  {name[:100]}")
