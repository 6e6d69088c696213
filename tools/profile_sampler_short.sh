#!/bin/bash
# rocprofv3 kernel-trace stats of a short sampler run (T = 40 reverse steps at the bench shape), GPU box, from the repo root:
#   bash tools/profile_sampler_short.sh <tag>
set -u
TAG=${1:-ps}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python3 $ROOT/bench.py --steps 1 --warmup 0 --timesteps 40 --no-cpu-baseline --no-train-probe > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log" || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
cd $ROOT
DB=$(find "$OUT/trace" -name '*.db' | head -1)
[ -n "$DB" ] && python3 tools/rocpd_stats.py "$DB" > "$OUT/kernel_stats.csv"
rm -rf "$OUT/trace"
head -9 "$OUT/kernel_stats.csv" | cut -c1-150
