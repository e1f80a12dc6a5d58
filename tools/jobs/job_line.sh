cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final; timeout 600 python bench.py > gpurun_out/final/bench_line.json 2> gpurun_out/final/bench_err.txt
python -c "
import json; d=json.load(open('gpurun_out/final/bench_line.json')); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['pcie_inclusive']['value'], d['gather']['frac'])"
