#!/bin/bash
# Bench + rocprofv3 kernel-trace stats + PMC passes (on the GPU box, from the repo root): bash tools/profile_bench.sh <tag>
set -u
TAG=${1:-r1}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench failed"; tail -5 "$OUT/bench.err"; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-train-probe > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log" || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
cd $ROOT
DB=$(find "$OUT/trace" -name '*.db' | head -1)
[ -n "$DB" ] && python3 tools/rocpd_stats.py "$DB" > "$OUT/kernel_stats.csv"
find "$OUT/trace" -name '*.db' -delete
bash tools/pmc_passes.sh $TAG/pmc || exit 1
python3 - "$OUT/pmc" > "$OUT/pmc_summary.txt" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'td_' in k: agg[k.split('(')[0][:100]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]): print(f"    {c:32s} {sum(agg[k][c])/len(agg[k][c]):.5g}  (n={len(agg[k][c])})")
PY
cat "$OUT/bench.json"; head -8 "$OUT/kernel_stats.csv" | cut -c1-160
