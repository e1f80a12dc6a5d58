cd $GRAFT_REPO_ROOT
for p in f32 bf16 fp8; do
timeout 300 python bench.py --mode sharded --precision $p --steps 400 --warmup 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sharded G=1 $p', round(d['value']/1e6,2), 'M inf/s', d['ms_per_step'])"
done
