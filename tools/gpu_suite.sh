#!/bin/bash
# One GPU-box call: the whole -m gpu suite, the WIP no-alloc test, then the driver's bench command.
# Every step has its own time limit; after anything but a clean pass / plain test failures nothing more runs.
set -u
tag=${1:-run}
out=gpurun_out/$tag
mkdir -p $out
step() {   # step <name> <seconds> <cmd...>
    local name=$1 secs=$2; shift 2
    echo "=== $name ($(date +%T))"
    timeout -k 10 $secs "$@" > $out/$name.log 2>&1
    local rc=$?
    echo "=== $name rc=$rc"
    tail -n 15 $out/$name.log
    return $rc
}
step tests 900 python3 -u -m pytest tests -m gpu -q --timeout 240 --timeout-method thread -rA -s
rc=$?
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping: tests ended with $rc"; exit $rc; fi
if [ -f tests/wip_no_alloc.py ]; then
    step noalloc 200 python3 -u -m pytest tests/wip_no_alloc.py -q --timeout 120 --timeout-method thread -rA
    rc2=$?
    if [ $rc2 -ne 0 ] && [ $rc2 -ne 1 ]; then echo "stopping: no-alloc test ended with $rc2"; exit $rc2; fi
fi
if [ "${2:-bench}" = "bench" ]; then
    step bench 580 python3 bench.py --gpus 1 --steps 20 --warmup 5
    rc3=$?
    tail -n 1 $out/bench.log > $out/bench.json
    [ $rc3 -ne 0 ] && exit $rc3
fi
exit $rc
