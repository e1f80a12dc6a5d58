cd $GRAFT_REPO_ROOT
for cfg in "f32 f32" "bf16 f32" "bf16 lp" "fp8 f32" "fp8 lp"; do
set -- $cfg
timeout 300 python bench.py --mode sharded --precision $1 --transport $2 --steps 400 --warmup 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sharded G=1 $1 transport=$2', round(d['value']/1e6,2), 'M inf/s', round(d['ms_per_step'],4), d['config']['pipelined_equals_stepwise'])"
done
