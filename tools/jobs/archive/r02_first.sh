# round 2, first GPU call: GPU test suite, smoke, the driver's own bench command, the default bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r02a/pytest_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02a/bench_driver_cmd.json 2> gpurun_out/r02a/bench_driver_cmd.err; tail -c 1500 gpurun_out/r02a/bench_driver_cmd.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02a/bench_driver_cmd.json'))
print('value',d['value'],'timed_s',d.get('timed_s'),'burst',d.get('burst',{}).get('value'))
print('roofline',d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('avg_launch_ms'))
print('groups',json.dumps(d.get('launch_group_table'),indent=0)[:1500])
print('pcie',d.get('pcie_inclusive',{}).get('value'), d.get('pcie_inclusive_streaming',{}).get('value'))
print('cpu',json.dumps(d.get('cpu_baseline'))[:1200])
for c in d.get('configs',[]): print('cfg',json.dumps(c)[:700])
print('gather',json.dumps(d.get('gather'))[:1200])
print('bank',json.dumps(d.get('gather_per_bank'))[:800])
PY
