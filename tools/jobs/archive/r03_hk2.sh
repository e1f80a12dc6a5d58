cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_hk2; mkdir -p $O
for nb in 16 64; do timeout -k 10 200 python tools/experiments/fused_hk_stamps.py $nb B 2>&1 | tail -22 | tee -a $O/stamps.txt; done
timeout -k 10 200 python tools/experiments/fused_hk_stamps.py 64 A 2>&1 | tail -22 | tee -a $O/stamps.txt
