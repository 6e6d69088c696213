#!/bin/bash
# PMC passes over the wide-hidden layer kernel (one pass per counter group): bash tools/pmc_wide.sh <tag>
set -u
TAG=${1:-pmcwide}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/wide_bench.py --hidden 512 --batch 8192 --seq-len 128 --timesteps 1 --reps 1"
timeout -s KILL 170 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d "$OUT/sq" -- $CMD > "$OUT/sq.log" 2>&1 || { echo "sq pass failed"; tail -5 "$OUT/sq.log"; exit 1; }
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$OUT/fetch.log"; exit 1; }
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1 || { echo "write pass failed"; tail -5 "$OUT/write.log"; exit 1; }
timeout -s KILL 170 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/tcc" -- $CMD > "$OUT/tcc.log" 2>&1 || { echo "tcc pass failed"; tail -5 "$OUT/tcc.log"; exit 1; }
cd $ROOT
python3 - "$OUT" > "$OUT/pmc_summary.txt" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'td_rnn_layer_wide' in k or 'td_head_wide' in k: agg[k.split('(')[0][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]): print(f"    {c:32s} {sum(agg[k][c])/len(agg[k][c]):.5g}  (n={len(agg[k][c])})")
PY
cat "$OUT/pmc_summary.txt"
