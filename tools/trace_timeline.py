#!/usr/bin/env python3
"""Kernel timeline of ONE steady-state iteration from a rocprofv3 --kernel-trace csv (two queues side by side):
usage: python tools/trace_timeline.py <kernel_trace.csv> [anchor-substring = k_rollout] [which occurrence from the end = 3]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_rollout"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
i0, i1 = idx[-back], idx[-back + 1]
t0 = int(rows[i0]["Start_Timestamp"])
qs = {}
print("# kernel                              queue  start      duration  end    gap-to-previous-on-queue  (us)")
last = {}
for r in rows[i0:i1 + 1]:
    q = qs.setdefault(r["Queue_Id"], "q%d" % (len(qs) + 1))
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    gap = s - last.get(q, s)
    last[q] = e
    print("%-36s %-4s %9.1f %9.1f %9.1f %7.1f" % (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:36], q, s, e - s, e, gap))
