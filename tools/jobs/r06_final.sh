#!/bin/bash
# round 6: what the driver runs at round end, on the final tree: the GPU suite, smoke(), the default bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r06_final_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r06_final_tests.log
[ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke()" || exit 1
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_final_bench.json 2> gpurun_out/r06_final_bench.err; rc=$?
tail -c 600 gpurun_out/r06_final_bench.json; tail -2 gpurun_out/r06_final_bench.err
exit $rc
