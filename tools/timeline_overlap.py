#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of the bench's timed loop and reports, for its steady state, what the streams did:
per queue busy time and the gaps between its consecutive dispatches, per kernel the time under the concurrent schedule, and how
many kernels ran at once, as fractions of the wall time.

usage: timeline_overlap.py <kernel_trace.csv> [skip_fraction=0.3] [keep_fraction=0.5]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
keep = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
fast = [r for r in rows if "fast_nms_k" in r["Kernel_Name"]]   # the window is taken inside the span of the extract kernels
t0, t1 = int(fast[0]["Start_Timestamp"]), int(fast[-1]["End_Timestamp"])
a, b = t0 + (t1 - t0) * skip, t0 + (t1 - t0) * (skip + keep)
sel = [r for r in rows if a <= int(r["Start_Timestamp"]) and int(r["End_Timestamp"]) <= b]
wall = (b - a) / 1e6
print(f"window {wall:.2f} ms, {len(sel)} dispatches")
byq = defaultdict(list)
for r in sel:
    byq[r["Queue_Id"]].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e6
    gaps = [(int(rs[i + 1]["Start_Timestamp"]) - int(rs[i]["End_Timestamp"])) / 1e3 for i in range(len(rs) - 1)]
    names = defaultdict(float)
    for r in rs:
        names[r["Kernel_Name"].split("(")[0].split("::")[-1]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    top = ", ".join(f"{k} {v / wall * 100:.0f}%" for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:6])
    big = sorted(gaps)[-5:] if gaps else []
    print(f"queue {q}: {len(rs)} dispatches, busy {busy / wall * 100:.0f}% of wall, gaps: median {sorted(gaps)[len(gaps) // 2] if gaps else 0:.1f} us, "
          f"sum {sum(gaps) / 1e3 / wall * 100:.0f}% of wall, largest {[round(g) for g in big]} us")
    print(f"    {top}")
# concurrency histogram
ev = []
for r in sel:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
hist, cur, last = defaultdict(float), 0, a
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
hist[cur] += b - last
print("kernels running at once: " + ", ".join(f"{k}: {v / 1e6 / wall * 100:.0f}%" for k, v in sorted(hist.items())))
