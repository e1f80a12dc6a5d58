cd $GRAFT_REPO_ROOT
for p in bf16 fp8; do
for st in 3 2; do
for t in "2 2" "4 2"; do
set -- $t
FR_LP_GEMM_STAGES=$st timeout 300 python bench.py --model C --batch 4096 --precision $p --threads $1 --depth $2 --no-cpu-baseline --no-model-c --steps 1000 --warmup 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C $p stages=$st threads $1 depth $2', round(d['value']/1e6,2), 'M inf/s')"
done
done
done
