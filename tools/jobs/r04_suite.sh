#!/bin/bash
# the whole GPU suite + smoke + the driver's bench command (its roofline objects quote the CSVs committed under profiles/)
set -o pipefail
O=gpurun_out/r04_suite; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SECONDS=0; timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $? wall $SECONDS s"
cp gpurun_out/bench_detail.json $O/bench_detail.json
python3 tools/check_evidence.py $O/bench_detail.json 2>&1 | tail -25
