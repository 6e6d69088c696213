#!/bin/bash
# quick rocprofv3 kernel stats of a short sampler run: bash tools/profile_short.sh <tag> [bench args]
set -u
TAG=${1:-short}; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python3 $ROOT/bench.py --timesteps 16 --probe-steps 2 --no-cpu-baseline --no-train-probe "$@" > "$OUT/bench.json" 2> "$OUT/trace.log" || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
cd $ROOT
DB=$(find "$OUT/trace" -name '*.db' | head -1)
[ -n "$DB" ] && python3 tools/rocpd_stats.py "$DB" > "$OUT/kernel_stats.csv"
find "$OUT/trace" -name '*.db' -delete
head -8 "$OUT/kernel_stats.csv" | cut -c1-150
