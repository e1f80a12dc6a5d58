cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2_f32clk
timeout 300 python tools/experiments/fused_stamps.py 32 200 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s2_f32clk/stamps.txt
