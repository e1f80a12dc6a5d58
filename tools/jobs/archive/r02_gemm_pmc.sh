cd $GRAFT_REPO_ROOT
bash tools/pmc_gemm.sh bf16 bf16_old FR_GEMM_PIPE=0 2>&1 | tail -4
bash tools/pmc_gemm.sh bf16 bf16_pipe5 FR_GEMM_PIPE=5 2>&1 | tail -4
