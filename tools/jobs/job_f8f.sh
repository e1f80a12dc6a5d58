cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fp8 or readme or random_custom" 2>&1 | tail -5
for cfg in "A 256" "B 1024"; do
set -- $cfg
timeout 300 python bench.py --model $1 --batch $2 --precision fp8 --no-cpu-baseline --no-model-c --steps 8000 --warmup 800 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg fp8', round(d['value']/1e6,2), 'M inf/s', d['config'].get('fc_tflops'))"
done
