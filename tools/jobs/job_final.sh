cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3
timeout 600 python bench.py > gpurun_out/final/bench_line.json 2> gpurun_out/final/bench_err.txt; tail -c 600 gpurun_out/final/bench_err.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/roof -- python3 $R/bench.py --roofline-only > $R/gpurun_out/final/roof.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/default -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/final/default.log 2>&1
cd $R
timeout 1500 bash tools/pmc_traffic.sh > gpurun_out/final/pmc.log 2>&1
tail -3 gpurun_out/final/pmc.log
python -c "
import json; d=json.load(open('gpurun_out/final/bench_line.json')); print(d['value'], d['roofline']['achieved'], d['roofline']['frac'], d.get('gather'))"
