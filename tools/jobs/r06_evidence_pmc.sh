#!/bin/bash
# round 6 evidence, part 2: the PMC passes of every roofline key (separate --pmc runs; tools/pmc_passes.sh) + the in-chain gather of the per-bank
# Model-C chains with the operand-type bank image on / off (read requests per launch: what the image is for).  PMC_KEYS / PMC_PART split the keys
# over several gpurun calls (a call is limited to 20 minutes); the parts are merged into profiles/r06_pmc.json afterwards.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_evidence
mkdir -p $O
PMC_NAME=r06_pmc.json bash tools/pmc_passes.sh ${PMC_KEYS:-} > $O/pmc_passes_${PMC_PART:-all}.log 2>&1; tail -1 $O/pmc_passes_${PMC_PART:-all}.log | cut -c1-300
cp gpurun_out/pmc/r06_pmc.json $O/r06_pmc_${PMC_PART:-all}.json
rm -rf gpurun_out/pmc
ls -la $O/r06_pmc_${PMC_PART:-all}.json
