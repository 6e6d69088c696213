#!/bin/bash
# rocprofv3 kernel-trace stats of training probes at given "batch ga steps" triples: bash tools/profile_train3.sh <tag> "8192 1 3" ...
set -u
TAG=${1:-train}; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
    set -- $cfg
    name=b$1h${4:-256}
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace_$name" -- python3 $ROOT/tools/train_probe.py --batch $1 --ga $2 --steps $3 --warmup 1 --hidden ${4:-256} > "$OUT/probe_$name.json" 2> "$OUT/trace_$name.log" || { echo "trace failed"; tail -5 "$OUT/trace_$name.log"; exit 1; }
    DB=$(find "$OUT/trace_$name" -name '*.db' | head -1)
    [ -n "$DB" ] && python3 $ROOT/tools/rocpd_stats.py "$DB" > "$OUT/kernel_stats_$name.csv"
    rm -rf "$OUT/trace_$name"
    cut -c1-200 "$OUT/probe_$name.json"
    head -22 "$OUT/kernel_stats_$name.csv" | cut -c1-150
done
