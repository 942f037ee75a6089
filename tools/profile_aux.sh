#!/bin/bash
# rocprofv3 passes over the kernels that bench.py does not launch: F-matrix RANSAC (tools/fmat_probe.py) and one LM iteration of
# 32 BA windows (tools/ba_probe.py).  Kernel stats + the SQ instruction/stall counters + HBM traffic, each in its own run.
set -e
OUT=${1:-gpurun_out/prof_aux}
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for probe in fmat ba; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/${probe}_stats" -- python3 "$REPO/tools/${probe}_probe.py" > "$REPO/$OUT/${probe}_stats.log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$REPO/$OUT/${probe}_pmc_sq" -- python3 "$REPO/tools/${probe}_probe.py" > "$REPO/$OUT/${probe}_pmc_sq.log" 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$REPO/$OUT/${probe}_pmc_fetch" -- python3 "$REPO/tools/${probe}_probe.py" > "$REPO/$OUT/${probe}_pmc_fetch.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$REPO/$OUT/${probe}_pmc_write" -- python3 "$REPO/tools/${probe}_probe.py" > "$REPO/$OUT/${probe}_pmc_write.log" 2>&1
done
