cd $GRAFT_REPO_ROOT
for mi in 16384 32768 65536; do
FR_FUSED_MAX_ITEMS=$mi timeout 600 python bench.py --model B --batch 1024 --precision bf16 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('max_items=$mi: B bf16 value %.1f M  launch %.1f us' % (d['value']/1e6, d['roofline']['avg_launch_ms']*1e3))"
done
