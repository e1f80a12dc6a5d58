cd $GRAFT_REPO_ROOT/tools/experiments
for prec in 1 2; do for ord in 1 3 4; do for prio in 0 1; do FR_GEMM_PRIO=$prio FR_GEMM_ORDER=$ord FR_GEMM_PIPE=15 ./gemm_pipe_bench $prec | tail -1 | sed "s/^/prio=$prio order=$ord /"; done; done; done
