cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "streaming or bf16 or driver" 2>&1 | tail -3
for cfg in "B 1024 f32" "B 1024 bf16" "A 256 bf16" "A 256 f32"; do
set -- $cfg
timeout 300 python bench.py --model $1 --batch $2 --precision $3 --no-cpu-baseline --no-model-c --steps 4000 --warmup 400 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['value']/1e6,2), 'M inf/s', d['config'].get('fc_tflops'))"
done
