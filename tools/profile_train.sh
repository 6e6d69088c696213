#!/bin/bash
# rocprofv3 kernel-trace stats of the training probe (on the GPU box, from the repo root): bash tools/profile_train.sh <tag>
set -u
TAG=${1:-train}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python3 $ROOT/bench.py --timesteps 8 --batch 2048 --probe-steps 1 --no-cpu-baseline --train-steps 20 > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log" || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
cd $ROOT
DB=$(find "$OUT/trace" -name '*.db' | head -1)
[ -n "$DB" ] && python3 tools/rocpd_stats.py "$DB" > "$OUT/kernel_stats.csv"
find "$OUT/trace" -name '*.db' -delete
head -30 "$OUT/kernel_stats.csv" | cut -c1-150
