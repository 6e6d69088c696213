#!/bin/bash
# PMC passes over a short bench run (one pass per counter group; see MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Usage (on the GPU box, from the repo root): bash tools/pmc_passes.sh <tag>
set -u
TAG=${1:-pmc}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --timesteps 4 --probe-steps 2 --no-cpu-baseline --no-train-probe --batch 16384"
timeout -s KILL 170 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_COEXEC_CYCLES \
    --output-format csv -d "$OUT/sq" -- $CMD > "$OUT/sq.log" 2>&1 || { echo "sq pass failed"; tail -5 "$OUT/sq.log"; exit 1; }
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$OUT/fetch.log"; exit 1; }
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1 || { echo "write pass failed"; tail -5 "$OUT/write.log"; exit 1; }
find "$OUT" -name '*.csv' | head; du -sh "$OUT"
