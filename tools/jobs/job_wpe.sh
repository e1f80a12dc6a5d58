cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "scores or streaming or known" 2>&1 | tail -3
for w in 2 4; do
echo "WPE=$w"
FR_FUSED_WPE=$w timeout 300 python bench.py --no-cpu-baseline --no-model-c 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'])"
FR_FUSED_WPE=$w timeout 300 python bench.py --no-cpu-baseline --no-model-c --threads 4 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('t4', d['value'])"
done
