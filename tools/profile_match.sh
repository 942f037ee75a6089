#!/bin/bash
# rocprofv3 passes over the matcher alone (tools/match_probe.py): kernel stats, SQ instruction / stall / MFMA counters, LDS.
set -e
OUT=${1:-gpurun_out/prof_match}
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/stats" -- python3 "$REPO/tools/match_probe.py" > "$REPO/$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$REPO/$OUT/pmc_sq" -- python3 "$REPO/tools/match_probe.py" > "$REPO/$OUT/pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d "$REPO/$OUT/pmc_mfma" -- python3 "$REPO/tools/match_probe.py" > "$REPO/$OUT/pmc_mfma.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$REPO/$OUT/pmc_fetch" -- python3 "$REPO/tools/match_probe.py" > "$REPO/$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$REPO/$OUT/pmc_write" -- python3 "$REPO/tools/match_probe.py" > "$REPO/$OUT/pmc_write.log" 2>&1
