mkdir -p gpurun_out
nproc > gpurun_out/host.txt; free -g >> gpurun_out/host.txt; rocm-smi --showmeminfo vram 2>/dev/null | head -8 >> gpurun_out/host.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -30 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -5 gpurun_out/smoke.log
timeout 600 python bench.py --steps 500 --warmup 50 > gpurun_out/bench1.log 2>&1; echo "bench exit $?" >> gpurun_out/bench1.log; tail -5 gpurun_out/bench1.log
