cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_hs2; mkdir -p $O
EXP=$GRAFT_REPO_ROOT/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
FR_LIB=$EXP FR_FUSED_ITEMS=65536 timeout -k 10 200 python tools/experiments/fused_hk_stamps.py 64 B 2>&1 | tail -40 | tee -a $O/stamps.txt
timeout -k 10 200 python tools/experiments/fused_hk_stamps.py 64 A 2>&1 | tail -40 | tee -a $O/stamps.txt
