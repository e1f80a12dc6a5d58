cd $GRAFT_REPO_ROOT
O=gpurun_out/s2_clk; mkdir -p $O
for r in 212 -2; do for nb in 16 1; do echo "=== knob $r, $nb batches per launch"; FR_FUSED_H_RING=$r timeout 300 python tools/experiments/fused_h_stamps.py $nb 2>&1 | grep -v amdgpu.ids | grep -v "^wave [1-35-7]"; done; done > $O/stamps.txt 2>&1
cat $O/stamps.txt
