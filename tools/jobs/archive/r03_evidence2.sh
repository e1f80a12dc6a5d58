# after r03_evidence.sh's CSVs are in profiles/: the driver's bench command again (its roofline objects quote `profiled_avg_launch_us` from the
# committed CSVs, which must be the ones of the shipped kernels), then the PMC passes of the persistent bf16 kernel
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_evidence2; mkdir -p $O
SECONDS=0; timeout -k 10 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench wall $SECONDS s"
bash tools/pmc_fused.sh B 1024 bf16 r03_B1024_bf16 > $O/pmc_fused.log 2>&1; tail -2 $O/pmc_fused.log | cut -c1-300
cp gpurun_out/pmc_fused/r03_B1024_bf16.json $O/ 2>/dev/null; ls $O
