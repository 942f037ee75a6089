#!/bin/bash
# frames/s of the default bench for the train-tile split of hamming_mfma_k (MSLAM_MATCH_SPLIT) and for the VALU matcher
for s in 1 2 4 6 8; do
  MSLAM_MATCH_SPLIT=$s python bench.py --no-parity --cpu-frames 0 --no-ingest > gpurun_out/ms_$s.json 2>/dev/null
  python - <<PY
import json; d=json.load(open("gpurun_out/ms_$s.json")); e=d["kernel_ms_per_frame_step_exclusive"]
print("split $s:", round(d["value"]), "frames/s; exclusive ms per 32 frames: mfma", e.get("hamming_mfma_k"), "expand", e.get("expand_pm1_k"), "finalize", e.get("knn2_finalize"))
PY
done
MSLAM_MATCH_MFMA=0 python bench.py --no-parity --cpu-frames 0 --no-ingest > gpurun_out/ms_valu.json 2>/dev/null
python - <<PY
import json; d=json.load(open("gpurun_out/ms_valu.json")); e=d["kernel_ms_per_frame_step_exclusive"]
print("VALU matcher:", round(d["value"]), "frames/s; exclusive: partial", e.get("hamming_knn2_partial"), "finalize", e.get("knn2_finalize"))
PY
