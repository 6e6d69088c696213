#!/bin/bash
# SQ counter passes over the training probe (one pass per group): bash tools/pmc_train.sh <tag> [batch] [ga]
set -u
TAG=${1:-pmctrain}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/train_probe.py --batch ${2:-2048} --ga ${3:-1} --steps 2 --warmup 1"
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/a" -- $CMD > "$OUT/a.log" 2>&1 || { echo "pass a failed"; tail -5 "$OUT/a.log"; exit 1; }
timeout -s KILL 170 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_FLAT \
    --output-format csv -d "$OUT/b" -- $CMD > "$OUT/b.log" 2>&1 || { echo "pass b failed"; tail -5 "$OUT/b.log"; }
cd $ROOT
python3 - "$OUT" > "$OUT/pmc_summary.txt" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'td_' in k: agg[k.split('(')[0][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]): print(f"    {c:32s} {sum(agg[k][c])/len(agg[k][c]):.5g}  (n={len(agg[k][c])})")
PY
grep -A16 "cluster" "$OUT/pmc_summary.txt" | head -80
