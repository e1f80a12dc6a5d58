#!/bin/bash
# round 6 evidence, part 3: the driver's own bench command with this round's summaries in profiles/, then tools/check_evidence.py
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_evidence
mkdir -p $O
SECONDS=0; timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $? wall $SECONDS s"; tail -c 400 $O/bench_line.err
cp gpurun_out/bench_detail.json $O/r06_bench_detail_driver_cmd.json; cp $O/bench_line.json $O/r06_bench_line_driver_cmd.json
python3 tools/check_evidence.py $O/r06_bench_detail_driver_cmd.json | tee $O/check_evidence.txt
