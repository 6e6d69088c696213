#!/bin/bash
# rocprofv3 kernel-trace stats of tools/config_bench.py (the MIMIC-III- and HiRID-shaped configurations): bash tools/profile_configs.sh <tag>
set -u
TAG=${1:-configs}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python3 $ROOT/tools/config_bench.py > "$OUT/config_bench.json" 2> "$OUT/trace.log" || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
DB=$(find "$OUT/trace" -name '*.db' | head -1)
[ -n "$DB" ] && python3 $ROOT/tools/rocpd_stats.py "$DB" > "$OUT/kernel_stats.csv"
rm -rf "$OUT/trace"
cut -c1-600 "$OUT/config_bench.json"
head -30 "$OUT/kernel_stats.csv" | cut -c1-160
