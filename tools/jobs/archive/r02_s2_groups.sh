cd $GRAFT_REPO_ROOT
O=gpurun_out/s2_groups; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "gather" 2>&1 | tail -4
timeout 900 python tools/experiments/gather_groups_sweep.py both > $O/sweep.txt 2>&1; cat $O/sweep.txt | grep -v amdgpu.ids
