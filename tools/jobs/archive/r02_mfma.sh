cd $GRAFT_REPO_ROOT/tools/experiments
for shape in 16 32; do for mode in 2 4 5 6; do ./mfma_bf16_rate $mode $shape | tail -1; done; done
