cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2_pmc
timeout 1100 bash tools/pmc_passes.sh > gpurun_out/s2_pmc/pmc.log 2>&1; tail -5 gpurun_out/s2_pmc/pmc.log | cut -c1-600
cp gpurun_out/pmc/r02_pmc.json gpurun_out/s2_pmc/r02_pmc.json
