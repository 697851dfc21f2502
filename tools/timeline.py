#!/usr/bin/env python3
"""Print one vector step's kernel timeline (start, end, duration in us; stream/queue) from a rocprofv3 results .db,
and the per-kernel averages over the run.  Usage: timeline.py <dir-or-db> [step-index]"""
import glob, os, sqlite3, sys
path = sys.argv[1]
db = path if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))[-1]
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end, queue_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "k_step" in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
i0, i1 = idx[k], idx[k + 1]
t0 = rows[i0][1]
print(f"step {k}: {(rows[i1][1] - t0) / 1000:.1f} us from k_step to k_step")
for r in rows[i0:i1 + 1]:
    print(f"{(r[1] - t0) / 1000:8.1f} {(r[2] - t0) / 1000:8.1f}  {(r[2] - r[1]) / 1000:6.1f}  q{r[3]}  {r[0][:90]}")
print()
for r in c.execute("select name, total_calls, average from top_kernels limit 16"):
    print(f"{r[1]:6d} x {float(r[2]):8.2f} us  {r[0][:100]}")
