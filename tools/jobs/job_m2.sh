cd $GRAFT_REPO_ROOT
FR_FUSED_M2=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "streaming or driver or known or concurrent" 2>&1 | tail -3
for cfg in "0 32" "1 64" "0 64"; do
set -- $cfg
FR_FUSED_M2=$1 FR_FUSED_GROUP=$2 timeout 300 python bench.py --no-cpu-baseline --no-model-c 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('m2=$1 group$2', round(d['value']/1e6,2), round(r['achieved'],1), r['avg_launch_ms'])"
done
