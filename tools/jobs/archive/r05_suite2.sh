#!/bin/bash
# round 5: what the driver runs at round end -- the GPU suite, smoke(), the bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tee gpurun_out/r05_suite2.log | tail -6 || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
