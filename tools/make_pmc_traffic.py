#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a tools/summarize_pmc.py summary: HBM bytes per launch of every kernel.

hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in KiB per dispatch; on gfx950 FETCH_SIZE
reads exactly half of the streamed bytes (MI355X_MICROARCH.md section HBM), confirmed for this library's 1/4/16 B-per-lane
loads with tools/calib/calib_fetch.hip (1 GiB read -> FETCH_SIZE 524299 KiB for all three widths); WRITE_SIZE is exact
(resize_k: 19.58 MB measured vs 19.52 MB of destination pixels per launch)."""
import json
import sys

summ = json.load(open(sys.argv[1]))
out = {"_batch": 32, "_source": "%s (tools/profile.sh + tools/summarize_pmc.py)" % sys.argv[1], "_formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes, averages per dispatch, bench default batch 32",
       "_calibration": {"read_1GiB_FETCH_SIZE_KiB": {"4B_per_lane": 524299.5, "16B_per_lane": 524298.5, "1B_per_lane_256MiB": 131082.0}}}
for k, v in summ.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v and not k.startswith("__amd"):
        out[k] = {"hbm_bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0, "FETCH_SIZE_KiB": v["FETCH_SIZE"],
                  "WRITE_SIZE_KiB": v["WRITE_SIZE"], "avg_ns": v.get("avg_ns"),
                  # wave-level instruction counts per launch (SQ_INSTS_*): the issue-side roofline of bench.py
                  "valu_insts_per_launch": v.get("SQ_INSTS_VALU"), "salu_insts_per_launch": v.get("SQ_INSTS_SALU"),
                  "lds_insts_per_launch": v.get("SQ_INSTS_LDS")}
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
print("wrote", sys.argv[2])
