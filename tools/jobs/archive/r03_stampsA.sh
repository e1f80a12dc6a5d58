# producer-side stamps of the persistent bf16 kernel on Model-A (256 batches per launch = 4 tiles per workgroup), diagnostic build
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_stampsA; mkdir -p $O
FR_LIB=$GRAFT_REPO_ROOT/gpu-fpga-recommendation-system_amd/libfleetrec_diag.so timeout -k 10 200 python tools/experiments/fused_hk_stamps.py 256 A 2>&1 | tail -36 | tee $O/stamps.txt
