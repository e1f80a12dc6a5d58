cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "tiled_gemm or bf16 or fp8 or scores or sharded or properties" 2>&1 | tail -3
for p in f32 bf16 fp8; do
timeout 300 python bench.py --model C --batch 4096 --precision $p --no-cpu-baseline --no-model-c --steps 1000 --warmup 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C 4096 $p', round(d['value']/1e6,2), 'M inf/s', d['config'].get('fc_tflops'))"
done
