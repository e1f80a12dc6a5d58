cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --model B --batch 1024 --precision f32 --no-cpu-baseline --no-model-c --steps 2000 --warmup 200 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B f32 fused', round(d['value']/1e6,2), 'M inf/s', d['config'].get('fc_tflops'))"
FR_FUSED=0 timeout 300 python bench.py --model B --batch 1024 --precision f32 --no-cpu-baseline --no-model-c --steps 2000 --warmup 200 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B f32 stage pipeline', round(d['value']/1e6,2), 'M inf/s', d['config'].get('fc_tflops'))"
