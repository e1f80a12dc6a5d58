cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for w in 2 4; do
FR_FUSED_WPE=$w timeout 300 python bench.py --no-cpu-baseline --no-model-c 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wpe$w', round(d['value']/1e6,2), round(d['roofline']['achieved'],1))"
done
done
