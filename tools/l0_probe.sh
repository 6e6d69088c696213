#!/bin/bash
# Per-kernel times of the denoiser forward at 16384 sequences with the unit-split family forced (is layer 0,
# which is bound by its gate math rather than MFMA, better off with two waves per SIMD?): bash tools/l0_probe.sh <tag>
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${1:-l0probe}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export TIMEDIFF_LAYER_KERNEL=${2:-u}
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python3 $ROOT/tools/perf_probe.py 16384 > "$OUT/probe.log" 2>&1
rc=$?
cd $ROOT
grep "fwd" "$OUT/probe.log"
[ $rc -ne 0 ] && { tail -5 "$OUT/probe.log"; exit $rc; }
DB=$(find "$OUT/trace" -name '*.db' | head -1)
[ -n "$DB" ] && python3 tools/rocpd_stats.py "$DB" > "$OUT/kernel_stats.csv"
find "$OUT/trace" -name '*.db' -delete
head -8 "$OUT/kernel_stats.csv" | cut -c1-170
