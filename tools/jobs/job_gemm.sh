cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "low_precision or bf16 or fp8" 2>&1 | tail -5
for p in bf16 fp8; do
for g in 0 -1; do
if [ $g = 0 ]; then export FR_LP_GEMM=0; else unset FR_LP_GEMM; fi
timeout 300 python bench.py --model C --batch 4096 --precision $p --no-cpu-baseline --no-model-c --steps 1000 --warmup 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C $p lp_gemm=$g', round(d['value']/1e6,2), 'M inf/s', d['config'].get('fc_tflops'))"
done
done
