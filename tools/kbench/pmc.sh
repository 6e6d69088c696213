#!/bin/bash
# usage: bash tools/kbench/pmc.sh <binary name in tools/kbench/build> [args...]   (on the GPU box)
set -u
BIN=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/kpmc_$BIN
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for pass in "sq1:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
            "sq2:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU" \
            "tcc1:FETCH_SIZE TCC_HIT_sum" "tcc2:WRITE_SIZE TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout -s KILL 60 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/$name" -- $ROOT/tools/kbench/build/$BIN "$@" > "$OUT/$name.log" 2>&1 || { echo "$name failed"; tail -3 "$OUT/$name.log"; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'td_rnn_layer' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg): print(f"{k:32s} {sum(agg[k])/len(agg[k]):.4g}  (n={len(agg[k])})")
PY
