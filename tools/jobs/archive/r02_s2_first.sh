# session 2 of round 2: state check on a fresh box (GPU suite, smoke, driver bench line) + vendor-library GEMM rates for reference
cd $GRAFT_REPO_ROOT
O=gpurun_out/s2_first; mkdir -p $O
timeout 300 python tools/experiments/blas_reference_rate.py $O/blas_reference_rate.json > $O/blas.log 2>&1; tail -20 $O/blas.log
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $O/pytest_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
SECONDS=0; timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench wall $SECONDS s"; tail -c 300 $O/bench_line.err
