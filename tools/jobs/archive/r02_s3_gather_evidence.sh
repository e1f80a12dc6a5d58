# PMC passes of the three gather keys with the software-pipelined kernel + rocprofv3 kernel stats of the two gather legs (one box)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s3_gev; mkdir -p $O $R/gpurun_out/pmc
cp profiles/r02_pmc.json gpurun_out/pmc/r02_pmc.json
timeout 1000 bash tools/pmc_passes.sh gather_C4096_per_table_uniform gather_C4096_per_table_zipf gather_C4096_per_bank_uniform > $O/pmc.log 2>&1; grep "^gather_" $O/pmc.log | cut -c1-600
cp gpurun_out/pmc/r02_pmc.json $O/r02_pmc.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bank -- python3 $R/bench.py --quick --no-gather-ab --legs bank > $O/bank_line.json 2> $O/bank.err &&
cp $(ls $O/stats_bank/*/*kernel_stats.csv | head -1) $O/gather_per_bank_uniform_kernel_stats.csv &&
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_table -- python3 $R/bench.py --quick --no-gather-ab --legs gather --gather-law uniform > $O/table_line.json 2> $O/table.err &&
cp $(ls $O/stats_table/*/*kernel_stats.csv | head -1) $O/gather_per_table_uniform_kernel_stats.csv
grep gather_pack $O/gather_per_bank_uniform_kernel_stats.csv $O/gather_per_table_uniform_kernel_stats.csv | cut -c1-260
