cd $GRAFT_REPO_ROOT
for p in bf16 fp8; do
for mt in 128 64 32; do
FR_LP_GEMM_MIN_TILES=$mt timeout 300 python bench.py --model C --batch 4096 --precision $p --threads 2 --depth 2 --no-cpu-baseline --no-model-c --steps 1000 --warmup 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C $p min_tiles=$mt', round(d['value']/1e6,2), 'M inf/s')"
done
done
