#!/bin/bash
# A-B-A-B timing of layer-kernel harness variants on ONE box (the in-kernel clock needs -DTD_STAMP builds).
#   bash tools/kbench/abab.sh <out tag> <binary> <binary> [...]      (args of the harness: env LB_ARGS, default "16384 272 5")
set -u
tag=$1; shift
out=gpurun_out/$tag.txt
mkdir -p gpurun_out
: > $out
for rep in 1 2; do
  for b in "$@"; do
    timeout -k 10 120 tools/kbench/bin/$b ${LB_ARGS:-16384 272 5} >> $out 2>&1 || { echo "$b failed"; tail -3 $out; exit 1; }
  done
done
cat $out
