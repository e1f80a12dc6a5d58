# session 2 evidence run: full GPU suite, smoke, the driver's bench command, rocprofv3 kernel stats of the roofline legs, PMC passes
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/pytest_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
SECONDS=0; timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench wall $SECONDS s"; tail -c 300 $O/bench_line.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_roofline -- python3 $R/bench.py --roofline-only > $O/stats_roofline.log 2>&1
echo roofline stats done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_gather_per_table_uniform -- python3 $R/bench.py --roofline-only --legs gather --gather-law uniform --no-gather-ab > $O/stats_gather_u.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_gather_per_bank_uniform -- python3 $R/bench.py --roofline-only --legs bank --no-gather-ab > $O/stats_gather_b.log 2>&1
echo gather stats done
for prec in f32 bf16 fp8; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_C4096_$prec -- python3 $R/bench.py --roofline-only --model C --batch 4096 --precision $prec > $O/stats_C4096_$prec.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_B1024_bf16 -- python3 $R/bench.py --roofline-only --model B --batch 1024 --precision bf16 > $O/stats_B1024_bf16.log 2>&1
echo config stats done
cd $R
for d in $O/stats_*/; do f=$(ls $d/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/$(basename $d)_kernel_stats.csv; done
bash tools/pmc_fused.sh B 1024 bf16 B1024_bf16 > $O/pmc_fused.log 2>&1; tail -2 $O/pmc_fused.log | cut -c1-1500
cp gpurun_out/pmc_fused/B1024_bf16.json $O/pmc_fused_B1024_bf16.json
ls $O
