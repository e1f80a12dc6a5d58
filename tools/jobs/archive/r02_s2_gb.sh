cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2_gb
timeout 600 python tools/experiments/gather_batch_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s2_gb/sweep.txt
timeout 300 python -m pytest tests/test_gpu_server.py -q -x 2>&1 | tail -2
