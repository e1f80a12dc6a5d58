# the fp8 form of the persistent kernel lives in the experiments library: its parity test runs there
cd $GRAFT_REPO_ROOT
FR_LIB=$GRAFT_REPO_ROOT/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so FR_FUSED_HK=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "fp8_persistent" 2>&1 | tail -3
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -x -k "fp8 or persistent or groups_above or dense_block" 2>&1 | tail -3
