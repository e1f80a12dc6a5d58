cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02_bench
SECONDS=0; timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench/line.json 2> gpurun_out/r02_bench/err.txt; echo "bench wall seconds: $SECONDS"; tail -c 300 gpurun_out/r02_bench/err.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02_bench/line.json'))
print('value',d['value'],'timed_s',d.get('timed_s'))
for c in d['configs']: print('cfg', c['workload'][:90], '|', c['dtype'], round(c['value']/1e6,1),'M kernel frac',round(c['roofline']['frac'],3), round(c['roofline']['avg_launch_ms']*1e3,1),'us')
print('gather',d['gather']['frac'],'bank',d['gather_per_bank']['frac'])
c=d['cpu_baseline']; print('cpu', c.get('gather_only'), c.get('fc_only'), c.get('end_to_end'), c.get('cores'))
PY
