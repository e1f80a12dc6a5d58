cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3_stream
timeout -k 10 600 python tools/experiments/gather_stream_batches.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s3_stream/batches.txt
