#!/bin/bash
# round 5 evidence, part 2: the PMC passes of every roofline key (separate --pmc runs; tools/pmc_passes.sh) + the CU-side groups of the persistent bf16 kernel
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_evidence
mkdir -p $O
PMC_NAME=r05_pmc.json bash tools/pmc_passes.sh > $O/pmc_passes.log 2>&1; tail -1 $O/pmc_passes.log | cut -c1-300
cp gpurun_out/pmc/r05_pmc.json $O/r05_pmc.json
rm -rf gpurun_out/pmc
ls -la $O/r05_pmc.json
