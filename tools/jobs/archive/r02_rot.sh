cd $GRAFT_REPO_ROOT
for rot in 0 1 0 1; do
FR_FUSED_ROT=$rot timeout 600 python bench.py --model B --batch 1024 --precision bf16 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rot=$rot B bf16 value %.1f M  launch %.1f us frac %.3f' % (d['value']/1e6, d['roofline']['avg_launch_ms']*1e3, d['roofline']['frac']))"
done
