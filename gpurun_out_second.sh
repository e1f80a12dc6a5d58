mkdir -p gpurun_out
for q in default 8 16 32; do
  echo "== GPU_MAX_HW_QUEUES=$q" >> gpurun_out/streams.log
  if [ $q = default ]; then ./tools/experiments/stream_concurrency 8 10000 >> gpurun_out/streams.log 2>&1; else GPU_MAX_HW_QUEUES=$q ./tools/experiments/stream_concurrency 8 10000 >> gpurun_out/streams.log 2>&1; fi
done
cat gpurun_out/streams.log
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1a -- python3 $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/bench_prof.log | cut -c1-600
find gpurun_out/prof_r1a -name "*stats*" | head
