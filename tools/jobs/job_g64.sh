cd $GRAFT_REPO_ROOT
for cfg in "2 32" "4 32" "4 64" "2 64"; do
set -- $cfg
FR_FUSED_WPE=$1 FR_FUSED_GROUP=$2 timeout 300 python bench.py --no-cpu-baseline --no-model-c 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('wpe$1 group$2', round(d['value']/1e6,2), round(r['achieved'],1), r['avg_launch_ms'])"
done
