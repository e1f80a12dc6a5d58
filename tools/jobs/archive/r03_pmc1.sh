# round 3: PMC passes of the wave-specialised bf16 fused kernel (Model-B 1024): traffic / L2 hit / MFMA busy, then the CU-side groups
cd $GRAFT_REPO_ROOT
bash tools/pmc_passes.sh fused_h_B1024_bf16 2>&1 | tail -3 | cut -c1-1500
cd $GRAFT_REPO_ROOT
bash tools/pmc_fused.sh B 1024 bf16 B1024_bf16_hs 2>&1 | tail -3 | cut -c1-2500
