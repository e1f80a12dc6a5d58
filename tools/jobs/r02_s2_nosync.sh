cd $GRAFT_REPO_ROOT
O=gpurun_out/s2_nosync; mkdir -p $O
for r in 16 -17; do echo "=== ring $r, 16 batches per launch"; FR_FUSED_H_RING=$r timeout 300 python tools/experiments/fused_h_stamps.py 16 2>&1 | grep -v amdgpu.ids; done > $O/w0.txt 2>&1
cat $O/w0.txt
