cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --legs value 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('value %.1f M  burst %.2f M (%.1f us per step)' % (d['value']/1e6, d['burst']['value']/1e6, d['burst']['ms_per_step']*1e3))"
FR_FUSED_M2=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --legs value 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('forced m2: value %.1f M  burst %.2f M (%.1f us per step)' % (d['value']/1e6, d['burst']['value']/1e6, d['burst']['ms_per_step']*1e3))"
done
