cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2_zc
for z in 0 1; do FR_SUBMIT_ZEROCOPY=$z timeout 300 python tools/experiments/submit_latency.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/s2_zc/lat.txt
