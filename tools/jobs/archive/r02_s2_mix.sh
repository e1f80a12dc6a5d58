cd $GRAFT_REPO_ROOT
O=gpurun_out/s2_mix; mkdir -p $O
for v in 0 1 2 3; do for g in 1 0; do for s in 0 1; do timeout 120 tools/experiments/fused_fc_mix $v $g $s 2>&1 | tail -2; done; done; done > $O/mix.txt 2>&1
cat $O/mix.txt
