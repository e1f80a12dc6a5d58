cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3_stream
FR_GATHER_STREAM=4 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "gather or tagged or blocked or shard" 2>&1 | tail -3 | tee gpurun_out/s3_stream/parity_stream4.txt &&
FR_GATHER_STREAM=2 FR_GATHER_ITEMS=2 FR_GATHER_STORE=16 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "gather or tagged or blocked or shard" 2>&1 | tail -3 | tee gpurun_out/s3_stream/parity_stream2.txt &&
timeout -k 10 600 python tools/experiments/gather_stream_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s3_stream/sweep.txt
