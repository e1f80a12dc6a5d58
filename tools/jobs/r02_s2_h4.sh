cd $GRAFT_REPO_ROOT
O=gpurun_out/s2_h4; mkdir -p $O
for r in 4 212 4 212; do echo "=== knob $r, 16 batches per launch"; FR_FUSED_H_RING=$r timeout 300 python tools/experiments/fused_h_stamps.py 16 2>&1 | grep -v amdgpu.ids | grep "knob\|span\|gather done\|FC3\|clock"; done > $O/stamps2.txt 2>&1
cat $O/stamps2.txt
