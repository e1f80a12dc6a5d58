cd $GRAFT_REPO_ROOT
for cfg in "B 1024 f32" "C 4096 f32"; do
set -- $cfg
timeout 300 python bench.py --model $1 --batch $2 --precision $3 --no-cpu-baseline --no-model-c --steps 2000 --warmup 200 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['value']/1e6,2), 'M inf/s', d['config'].get('fc_tflops'))"
done
