#!/bin/bash
# rocprofv3 passes over one LM iteration's device work (tools/ba_probe.py): kernel stats, SQ instruction / wait counters.
set -e
OUT=${1:-gpurun_out/prof_ba}
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/stats" -- python3 "$REPO/tools/ba_probe.py" > "$REPO/$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$REPO/$OUT/pmc_sq" -- python3 "$REPO/tools/ba_probe.py" > "$REPO/$OUT/pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$REPO/$OUT/pmc_lds" -- python3 "$REPO/tools/ba_probe.py" > "$REPO/$OUT/pmc_lds.log" 2>&1
cd "$REPO" && python3 tools/summarize_pmc.py "$OUT" > "$OUT/summary.json"
