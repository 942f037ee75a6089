#!/usr/bin/env python3
"""Per-dispatch timeline of one batched extract+match+PnP step from a rocprofv3 --kernel-trace CSV:
kernel, grid, duration and the gap to the previous dispatch on the same queue (launch/drain overhead)."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
byq = defaultdict(list)
for r in rows:
    byq[r.get("Queue_Id", "0")].append(r)
for q, rs in byq.items():
    # last complete step: take the last 40 dispatches
    tail = rs[-int(sys.argv[2]) if len(sys.argv) > 2 else -40:]
    print(f"queue {q}: {len(rs)} dispatches")
    prev_end = None
    for r in tail:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        name = r["Kernel_Name"].split("(")[0][-28:]
        print(f"  {name:28s} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>9s} dur {(e - s) / 1e3:8.1f} us  gap {gap:7.1f} us")
        prev_end = e
