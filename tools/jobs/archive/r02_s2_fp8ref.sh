cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2_fp8ref
timeout 300 python tools/experiments/blas_reference_rate.py gpurun_out/s2_fp8ref/blas_reference_rate.json 2>&1 | grep -v amdgpu.ids | tail -22
