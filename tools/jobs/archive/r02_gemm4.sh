cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tiled_gemm or sharded" 2>&1 | tail -4
for rep in 1 2; do for ns in 0 15; do
    FR_GEMM_PIPE=$ns timeout 600 python bench.py --model C --batch 4096 --precision bf16 > gpurun_out/r02h/c_bf16_ns${ns}_$rep.json 2>/dev/null
    python - <<PY
import json
d=json.load(open('gpurun_out/r02h/c_bf16_ns${ns}_$rep.json'))
print('rep $rep PIPE=$ns bf16 value %.1f M  layers(us) %s  FC1 frac %.3f' % (d['value']/1e6, [round(1e3*x,1) for x in d['layer_launch_ms']], d['roofline']['frac']))
PY
done; done
