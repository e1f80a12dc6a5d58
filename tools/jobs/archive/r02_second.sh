# round 2, second GPU call: whole GPU suite (no -x), gather sweep, PMC passes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee gpurun_out/r02b/pytest_tail.txt
timeout 600 python tools/experiments/gather_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02b/gather_sweep.txt
timeout 1500 bash tools/pmc_passes.sh > gpurun_out/r02b/pmc.log 2>&1; tail -5 gpurun_out/r02b/pmc.log
