#!/bin/bash
# round 5: where the host-fed stream loses its 5 %: kernel + memory-copy timeline of both request streams
set -o pipefail
mkdir -p gpurun_out/r05_hostfed
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/host_fed_run.py 4 2 0.5 2>&1 | tee $R/gpurun_out/r05_hostfed/plain_4x2.txt &&
python3 $R/tools/host_fed_run.py 2 2 0.5 2>&1 | tee $R/gpurun_out/r05_hostfed/plain_2x2.txt &&
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r05_hostfed/trace -- python3 $R/tools/host_fed_run.py 4 2 0.25 2>&1 | tail -5 | tee $R/gpurun_out/r05_hostfed/traced_4x2.txt
ls -R $R/gpurun_out/r05_hostfed/trace | head -20
for f in $(find $R/gpurun_out/r05_hostfed/trace -name "*memory_copy_trace.csv"); do head -3 $f; wc -l $f; done
for f in $(find $R/gpurun_out/r05_hostfed/trace -name "*kernel_trace.csv"); do head -2 $f; wc -l $f; done
