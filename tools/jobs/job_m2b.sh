cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-model-c 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('default', round(d['value']/1e6,2), round(r['achieved'],1), r['avg_launch_ms'], r['kernel'][:40])"
done
timeout 300 python bench.py --no-cpu-baseline --no-model-c --threads 4 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('threads 4', round(d['value']/1e6,2))"
timeout 300 python bench.py --no-cpu-baseline --no-model-c --threads 1 --depth 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('threads 1 depth 2', round(d['value']/1e6,2))"
