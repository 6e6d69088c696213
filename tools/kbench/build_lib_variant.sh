#!/bin/bash
# A diagnostic variant of the library: ONE source rebuilt with extra -D flags, linked with the product's other objects.
#   bash tools/kbench/build_lib_variant.sh <name> <source.hip> [-D...]   ->  tools/kbench/bin/lib<name>.so   (use: TIMEDIFF_HIP_LIB=...)
set -eu
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/../.." && pwd)
cs=$root/timediff_amd/csrc
mkdir -p $root/tools/kbench/bin $cs/build/variant
make -C $cs -j8 > /dev/null
obj=$cs/build/variant/${name}_$(basename $src .hip).o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wno-pass-failed -Wno-unused-value "$@" -c $cs/$src -o $obj
others=$(ls $cs/build/*.o | grep -v "/$(basename $src .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/kbench/bin/lib$name.so $others $obj
echo tools/kbench/bin/lib$name.so
