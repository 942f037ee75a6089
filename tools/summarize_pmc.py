#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output of tools/profile.sh: per-kernel mean of every counter (+ kernel stats)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out = {}
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        name = k.replace("(anonymous namespace)::", "").split("(")[0].strip()
        for c, v in d.items():
            out.setdefault(name, {})[c] = sum(v) / len(v)
            out[name]["dispatches"] = len(v)
for f in glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].strip()
        out.setdefault(name, {}).update(avg_ns=float(r["AverageNs"]), calls=int(r["Calls"]), pct=float(r["Percentage"]))
print(json.dumps(out, indent=1, sort_keys=True))
