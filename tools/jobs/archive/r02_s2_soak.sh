cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2_soak
timeout 300 python tools/experiments/soak.py 40 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s2_soak/soak.txt
