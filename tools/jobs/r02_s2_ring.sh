cd $GRAFT_REPO_ROOT
O=gpurun_out/s2_ring; mkdir -p $O
for r in 16 8 12 20 24 32; do echo "=== ring $r, 16 batches per launch"; FR_FUSED_H_RING=$r timeout 300 python tools/experiments/fused_h_stamps.py 16 2>&1 | grep -v amdgpu.ids; done > $O/ring.txt 2>&1
for nb in 1 2 4 8; do echo "=== ring 16, $nb batches per launch"; timeout 300 python tools/experiments/fused_h_stamps.py $nb 2>&1 | grep -v amdgpu.ids; done > $O/nb.txt 2>&1
cat $O/ring.txt $O/nb.txt
