#!/bin/bash
# Builds one variant of the layer-kernel timing harness (here, on the CPU; the binary travels with the tree).
#   bash tools/kbench/build.sh <name> [-D...]      ->  tools/kbench/build/<name>
set -eu
name=$1; shift
cd "$(dirname "$0")"
mkdir -p build bin
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-pass-failed -Wno-unused-value "$@" ${SRC:-layer_bench.hip} -o bin/$name
