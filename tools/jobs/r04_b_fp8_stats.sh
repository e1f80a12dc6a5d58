#!/bin/bash
# kernel stats of the Model-B fp8 row's own command (the row was added to the bench line late in the round), then the bench line again
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_b_fp8_stats; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -- python3 $R/bench.py --roofline-only --model B --batch 1024 --precision fp8 > $O/stats.log 2>&1
f=$(ls $O/s/*/*kernel_stats.csv | head -1); cp $f $O/B1024_fp8_kernel_stats.csv; cp $f $R/profiles/r04_B1024_fp8_kernel_stats.csv; rm -rf $O/s
cd $R
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $?"
cp gpurun_out/bench_detail.json $O/bench_detail.json
python3 tools/check_evidence.py $O/bench_detail.json 2>&1 | tail -16
