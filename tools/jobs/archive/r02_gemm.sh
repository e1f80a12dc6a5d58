cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tiled_gemm or bf16_chain or fp8_chain" 2>&1 | tail -15
for ns in 0 4 5 6; do
  for prec in bf16 fp8; do
    FR_GEMM_PIPE=$ns timeout 600 python bench.py --model C --batch 4096 --precision $prec > gpurun_out/r02e/c_${prec}_ns$ns.json 2>/dev/null
    python - <<PY
import json
d=json.load(open('gpurun_out/r02e/c_${prec}_ns$ns.json'))
print('NS=$ns $prec value %.1f M  layers(us) %s  FC1 frac %.3f' % (d['value']/1e6, [round(1e3*x,1) for x in d['layer_launch_ms']], d['roofline']['frac']))
PY
  done
done
