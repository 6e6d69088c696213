#!/bin/bash
# The library with the cluster kernels' phase stamps compiled in (-DTDC_TRACE): tools/kbench/bin/libtrace.so
set -eu
root=$(cd "$(dirname "$0")/../.." && pwd)
cs=$root/timediff_amd/csrc
mkdir -p $root/tools/kbench/bin $cs/build/variant
make -C $cs -j8 > /dev/null
for f in rnn_cluster rnn_cluster_bwd; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wno-pass-failed -DTDC_TRACE "$@" -c $cs/$f.hip -o $cs/build/variant/trace_$f.o &
done
wait
others=$(ls $cs/build/*.o | grep -v "/rnn_cluster.o\|/rnn_cluster_bwd.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/kbench/bin/libtrace.so $others $cs/build/variant/trace_rnn_cluster.o $cs/build/variant/trace_rnn_cluster_bwd.o
echo tools/kbench/bin/libtrace.so
