#!/bin/bash
# rocprofv3 passes over the bench command (run on the GPU box through gpurun, from the repo root):
#   1. --kernel-trace --stats           per-kernel time
#   2. --pmc SQ issue/stall counters    where the dominant kernels spend their cycles
#   3. --pmc FETCH_SIZE, 4. --pmc WRITE_SIZE   HBM-side traffic (separate passes: TCC slots)
# Counters are collected in their own runs (never combined with trace domains other than kernel-trace).
set -e
OUT=${1:-gpurun_out/prof}
shift || true
ARGS=${@:---steps 3 --warmup 1 --seq-len 16 --cpu-frames 0 --no-ingest --no-parity --no-ba}
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export BENCH_NO_F32_LEG=1   # the headline configuration only (PnP inlier test in double precision): no second leg with pnp_score32_k in the averages
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/stats" -- python3 "$REPO/bench.py" $ARGS > "$REPO/$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$REPO/$OUT/pmc_sq" -- python3 "$REPO/bench.py" $ARGS > "$REPO/$OUT/pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d "$REPO/$OUT/pmc_sq2" -- python3 "$REPO/bench.py" $ARGS > "$REPO/$OUT/pmc_sq2.log" 2>&1 || true
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$REPO/$OUT/pmc_fetch" -- python3 "$REPO/bench.py" $ARGS > "$REPO/$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$REPO/$OUT/pmc_write" -- python3 "$REPO/bench.py" $ARGS > "$REPO/$OUT/pmc_write.log" 2>&1
find "$REPO/$OUT" -name "*.csv" | head -30
