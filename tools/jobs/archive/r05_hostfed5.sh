#!/bin/bash
# round 5: kernel + memory-copy timelines of the host-fed stream, round 4's commands (FR_HOST_ZEROCOPY=0, experiments build) and this round's form (product build),
# reduced by tools/trace_host_fed.py -> profiles/r05_host_fed_timeline.txt
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_hostfed
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FR_LIB=$R/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so FR_HOST_ZEROCOPY=0 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace_r04form -- python3 $R/tools/host_fed_run.py 4 2 0.3 2>&1 | grep -v "^[EW]2" | tee $O/trace_r04form.txt &&
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace_r05form -- python3 $R/tools/host_fed_run.py 4 1 0.3 2>&1 | grep -v "^[EW]2" | tee $O/trace_r05form.txt &&
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace_r05form42 -- python3 $R/tools/host_fed_run.py 4 2 0.3 2>&1 | grep -v "^[EW]2" | tee $O/trace_r05form42.txt
for f in r04form r05form r05form42; do
  echo "######## $f: $(grep -h 'host_fed /' $O/trace_$f.txt)"
  python3 $R/tools/trace_host_fed.py $(dirname $(find $O/trace_$f -name "*kernel_trace.csv" | head -1))
done > $O/timeline_reduced.txt 2>&1
rm -rf $O/trace_r04form $O/trace_r05form $O/trace_r05form42 $O/trace
cat $O/timeline_reduced.txt
