cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16" 2>&1 | tail -5
for m in A B; do
b=256; [ $m = B ] && b=1024
timeout 300 python bench.py --model $m --batch $b --precision bf16 --no-cpu-baseline --no-model-c --steps 4000 --warmup 400 2>&1 | tail -1 | cut -c1-400
FR_FUSED=0 timeout 300 python bench.py --model $m --batch $b --precision bf16 --no-cpu-baseline --no-model-c --steps 4000 --warmup 400 2>&1 | tail -1 | cut -c1-300
done
