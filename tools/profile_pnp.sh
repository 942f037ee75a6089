#!/bin/bash
# rocprofv3 passes over the PnP RANSAC alone (tools/pnp_probe.py): kernel stats, SQ instruction / stall / MFMA counters.
set -e
OUT=${1:-gpurun_out/prof_pnp}
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/stats" -- python3 "$REPO/tools/pnp_probe.py" > "$REPO/$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$REPO/$OUT/pmc_sq" -- python3 "$REPO/tools/pnp_probe.py" > "$REPO/$OUT/pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d "$REPO/$OUT/pmc_mfma" -- python3 "$REPO/tools/pnp_probe.py" > "$REPO/$OUT/pmc_mfma.log" 2>&1
