#!/bin/bash
# rocprofv3 kernel-trace stats of the wide-hidden workloads (on the GPU box, from the repo root): bash tools/profile_wide.sh <tag>
set -u
TAG=${1:-wide}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python3 $ROOT/tools/wide_bench.py --hidden 1024 --batch 8192 --seq-len 128 --train-batch 512 > "$OUT/wide_under_rocprof.json" 2> "$OUT/trace.log" || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
cd $ROOT
DB=$(find "$OUT/trace" -name '*.db' | head -1)
[ -n "$DB" ] && python3 tools/rocpd_stats.py "$DB" > "$OUT/kernel_stats.csv"
find "$OUT/trace" -name '*.db' -delete
head -24 "$OUT/kernel_stats.csv" | cut -c1-170
