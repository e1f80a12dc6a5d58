#!/bin/bash
# round 6: the whole GPU suite after the fr_comm transport split / chain-width contract / per-item tolerance, then the default bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/r06_suite1.log 2>&1
rc=$?
tail -25 gpurun_out/r06_suite1.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py > gpurun_out/r06_bench1.json 2> gpurun_out/r06_bench1.err
rc=$?
tail -c 3000 gpurun_out/r06_bench1.json; tail -5 gpurun_out/r06_bench1.err
exit $rc
