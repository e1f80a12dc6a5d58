cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16 or fp8 or gather or scores or properties" 2>&1 | tail -3
for t in "2 2" "2 4" "4 2"; do
set -- $t
timeout 300 python bench.py --model C --batch 4096 --precision bf16 --threads $1 --depth $2 --no-cpu-baseline --no-model-c --steps 1000 --warmup 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C bf16 threads $1 depth $2', round(d['value']/1e6,2), 'M inf/s', d['config'].get('fc_tflops'))"
done
