# round 3 evidence run: the driver's bench command, rocprofv3 --kernel-trace --stats of every roofline leg's own command (single stream)
# and of the DEFAULT four-stream headline, then the PMC passes of every key.  Copies of the summaries go to profiles/r03_*.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_evidence
mkdir -p $O
SECONDS=0; timeout -k 10 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench wall $SECONDS s"; tail -c 200 $O/bench_line.err
cd /tmp && export TMPDIR=/tmp
st() { # name, bench args...
  n=$1; shift
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$n -- python3 $R/bench.py "$@" > $O/stats_$n.log 2>&1 || echo "stats $n failed"
  f=$(ls $O/stats_$n/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/${n}_kernel_stats.csv
}
st roofline --roofline-only
st value_4streams --legs none --quick
st gather_per_table_uniform --roofline-only --legs gather --gather-law uniform --no-gather-ab
st gather_per_table_zipf --roofline-only --legs gather --gather-law zipf --no-gather-ab
st gather_per_bank_uniform --roofline-only --legs bank --no-gather-ab
for prec in f32 bf16 fp8; do st C4096_$prec --roofline-only --model C --batch 4096 --precision $prec; done
st B1024_bf16 --roofline-only --model B --batch 1024 --precision bf16
st B1024_f32 --roofline-only --model B --batch 1024 --precision f32
echo stats done
cd $R
bash tools/pmc_passes.sh > $O/pmc_passes.log 2>&1; tail -1 $O/pmc_passes.log | cut -c1-300
cp gpurun_out/pmc/r03_pmc.json $O/r03_pmc.json
ls $O | head -40
