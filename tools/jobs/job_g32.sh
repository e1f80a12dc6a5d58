cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "tiled_gemm or properties or scores" 2>&1 | tail -3
for g in 0 -1; do
if [ $g = 0 ]; then export FR_LP_GEMM=0; else unset FR_LP_GEMM; fi
timeout 300 python bench.py --model C --batch 4096 --precision f32 --no-cpu-baseline --no-model-c --steps 300 --warmup 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C f32 lp_gemm=$g', round(d['value']/1e6,2), 'M inf/s', d['config'].get('fc_tflops'))"
done
