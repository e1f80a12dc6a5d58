#!/bin/bash
# round 4 evidence run, one box, one job: rocprofv3 --kernel-trace --stats of every roofline leg's own command (single stream) and of the
# DEFAULT four-stream headline, the PMC passes of every key, the CSVs copied into the box's profiles/ (the roofline objects quote them),
# then the driver's bench command and the CU-side PMC groups of the persistent bf16 kernel.  Copies of the summaries go to profiles/r04_*.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_evidence
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
st() { # name, bench args...
  n=$1; shift
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$n -- python3 $R/bench.py "$@" > $O/stats_$n.log 2>&1 || echo "stats $n failed"
  f=$(ls $O/stats_$n/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/${n}_kernel_stats.csv
  rm -rf $O/stats_$n
  echo "stats $n done"
}
st roofline --roofline-only
st value_4streams --legs none --quick
st gather_per_table_uniform --roofline-only --legs gather --gather-law uniform --no-gather-ab
st gather_per_table_zipf --roofline-only --legs gather --gather-law zipf --no-gather-ab
st gather_per_bank_uniform --roofline-only --legs bank --no-gather-ab
for prec in f32 bf16 fp8; do st C4096_$prec --roofline-only --model C --batch 4096 --precision $prec; done
st B1024_bf16 --roofline-only --model B --batch 1024 --precision bf16
st B1024_bf16_per_bank --roofline-only --model B --batch 1024 --precision bf16 --per-bank
st B1024_f32 --roofline-only --model B --batch 1024 --precision f32
st A256_bf16 --roofline-only --model A --batch 256 --precision bf16
st A256_fp8 --roofline-only --model A --batch 256 --precision fp8
cd $R
PMC_NAME=r04_pmc.json bash tools/pmc_passes.sh > $O/pmc_passes.log 2>&1; tail -1 $O/pmc_passes.log | cut -c1-300
cp gpurun_out/pmc/r04_pmc.json $O/r04_pmc.json
for f in $O/*_kernel_stats.csv; do cp $f $R/profiles/r04_$(basename $f); done
cp $O/r04_pmc.json $R/profiles/r04_pmc.json
SECONDS=0; timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $? wall $SECONDS s"; tail -c 200 $O/bench_line.err
cp gpurun_out/bench_detail.json $O/bench_detail.json
bash tools/pmc_fused.sh B 1024 bf16 r04_B1024_bf16 > $O/pmc_fused.log 2>&1; tail -1 $O/pmc_fused.log | cut -c1-300
cp gpurun_out/pmc_fused/r04_B1024_bf16.json $O/ 2>/dev/null
rm -rf gpurun_out/pmc gpurun_out/pmc_fused/r04_B1024_bf16
ls $O | head -50
