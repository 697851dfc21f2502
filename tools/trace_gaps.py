#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace csv: per queue, the idle time between consecutive kernels (start of the next - end of the
previous), grouped by the pair of kernels, and each kernel's duration.  usage: trace_gaps.py <kernel_trace.csv> [skip_fraction]"""
import collections, csv, re, statistics, sys


def short(n):
    m = re.search(r"(k_\w+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:40]


rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * skip):]
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
for q, ks in sorted(byq.items()):
    gaps, durs = collections.defaultdict(list), collections.defaultdict(list)
    for a, b in zip(ks, ks[1:]):
        gaps[(a[2], b[2])].append((b[0] - a[1]) / 1e3)
    for k in ks:
        durs[k[2]].append((k[1] - k[0]) / 1e3)
    print(f"queue {q}: {len(ks)} kernels")
    for k, v in sorted(durs.items(), key=lambda kv: -len(kv[1])):
        if len(v) >= 5:
            print(f"   {k:36s} x{len(v):6d}  median {statistics.median(v):7.2f} us")
    for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
        if len(v) >= 5:
            print(f"   gap {k[0]:30s} -> {k[1]:30s} x{len(v):6d}  median {statistics.median(v):7.2f}  p10 {sorted(v)[len(v) // 10]:7.2f} us")
if len(sys.argv) > 3:      # a stretch of the busiest queue, kernel by kernel
    q = max(byq, key=lambda k: len(byq[k]))
    ks = byq[q][len(byq[q]) // 2:][:int(sys.argv[3])]
    t0 = ks[0][0]
    for a in ks:
        print(f"   {(a[0] - t0) / 1e3:9.2f} {(a[1] - t0) / 1e3:9.2f}  {(a[1] - a[0]) / 1e3:7.2f}  {a[2]}")
