# the whole evidence of the round in one job: r03_evidence.sh (kernel-stat CSVs of every roofline leg's own command, PMC passes), the CSVs
# copied into the box's profiles/, then the driver's bench command (its roofline objects quote the CSVs) and the PMC passes of the bf16 kernel
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
bash tools/jobs/r03_evidence.sh > gpurun_out/r03_evidence_stage1.log 2>&1; tail -3 gpurun_out/r03_evidence_stage1.log
O=$R/gpurun_out/r03_evidence
for f in $O/*_kernel_stats.csv; do cp $f $R/profiles/r03_$(basename $f); done
cp $O/r03_pmc.json $R/profiles/r03_pmc.json
O2=$R/gpurun_out/r03_evidence2; mkdir -p $O2
cd $R
SECONDS=0; timeout -k 10 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O2/bench_line.json 2> $O2/bench_line.err; echo "bench wall $SECONDS s"
bash tools/pmc_fused.sh B 1024 bf16 r03_B1024_bf16 > $O2/pmc_fused.log 2>&1; tail -1 $O2/pmc_fused.log | cut -c1-200
cp gpurun_out/pmc_fused/r03_B1024_bf16.json $O2/ 2>/dev/null; ls $O2
