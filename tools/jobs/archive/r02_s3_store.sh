cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3_store
timeout -k 10 240 tools/experiments/copy_ceiling 2>&1 | tee gpurun_out/s3_store/copy_ceiling.txt &&
timeout -k 10 600 python tools/experiments/gather_store_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s3_store/store_sweep.txt
