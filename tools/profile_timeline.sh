#!/bin/bash
# Kernel timeline of one steady-state optimizer step: bash tools/profile_timeline.sh <tag> <batch> <ga> [shape] [back]
set -u
TAG=${1:-tl}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT/trace" -- python3 $ROOT/tools/train_probe.py --batch ${2:-32} --ga ${3:-2} --shape ${4:-eicu} --steps 12 --warmup 5 > "$OUT/probe.json" 2> "$OUT/trace.log" || { tail -5 "$OUT/trace.log"; exit 1; }
DB=$(find "$OUT/trace" -name '*.db' | head -1)
python3 $ROOT/tools/rocpd_timeline.py "$DB" ${5:-2} > "$OUT/timeline.txt"
rm -rf "$OUT/trace"
cat "$OUT/timeline.txt"
