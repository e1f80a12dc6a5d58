cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2_pmc gpurun_out/pmc
cp profiles/r02_pmc.json gpurun_out/pmc/r02_pmc.json
timeout 1100 bash tools/pmc_passes.sh "$@" > gpurun_out/s2_pmc/pmc2.log 2>&1; grep "^fused_\|^gemm_C\|^gather_" gpurun_out/s2_pmc/pmc2.log | cut -c1-700
cp gpurun_out/pmc/r02_pmc.json gpurun_out/s2_pmc/r02_pmc.json
