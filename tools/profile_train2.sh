#!/bin/bash
# rocprofv3 kernel-trace stats of the two training probes, one profile each: bash tools/profile_train2.sh <tag>
set -u
TAG=${1:-train}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in "32 2 20" "2048 1 6"; do
    set -- $cfg
    name=b$1
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace_$name" -- python3 $ROOT/tools/train_probe.py --batch $1 --ga $2 --steps $3 > "$OUT/probe_$name.json" 2> "$OUT/trace_$name.log" || { echo "trace failed"; tail -5 "$OUT/trace_$name.log"; exit 1; }
    DB=$(find "$OUT/trace_$name" -name '*.db' | head -1)
    [ -n "$DB" ] && python3 $ROOT/tools/rocpd_stats.py "$DB" > "$OUT/kernel_stats_$name.csv"
    rm -rf "$OUT/trace_$name"
    cat "$OUT/probe_$name.json"
    head -16 "$OUT/kernel_stats_$name.csv" | cut -c1-170
done
