cd $GRAFT_REPO_ROOT/tools/experiments
for rep in 1 2; do for v in 0 1; do FR_GEMM_F8_16=$v ./gemm_pipe_bench 2 | tail -1 | sed "s/^/f8_16=$v /"; done; done
cd $GRAFT_REPO_ROOT
FR_GEMM_F8_16=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tiled_gemm" 2>&1 | grep -E "passed|failed" | tail -2
for v in 0 1; do FR_GEMM_F8_16=$v timeout 600 python bench.py --model C --batch 4096 --precision fp8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('f8_16=$v fp8 value %.2f M  layers(us) %s' % (d['value']/1e6, [round(1e3*x,1) for x in d['layer_launch_ms']]))"; done
