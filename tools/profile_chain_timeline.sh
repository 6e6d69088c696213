#!/bin/bash
# One reverse step of the sampler as a kernel timeline: bash tools/profile_chain_timeline.sh <tag> <mimiciii|hirid> <batch>
set -u
TAG=${1:-chain}; SHAPE=${2:-mimiciii}; B=${3:-256}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT/trace" -- python3 $ROOT/tools/chain_probe.py $SHAPE $B ${4:-64} > "$OUT/probe.log" 2> "$OUT/trace.log" || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
DB=$(find "$OUT/trace" -name '*.db' | head -1)
[ -n "$DB" ] && python3 $ROOT/tools/chain_timeline.py "$DB" > "$OUT/timeline_${SHAPE}_$B.txt"
rm -rf "$OUT/trace"
cat "$OUT/probe.log" "$OUT/timeline_${SHAPE}_$B.txt"
