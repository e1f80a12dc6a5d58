cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
rocprofv3 -L > gpurun_out/r02g/counters.txt 2>&1
for ab in 0 1 2; do
  for prec in bf16 fp8; do
    FR_GEMM_ABLATE=$ab timeout 600 python bench.py --model C --batch 4096 --precision $prec --quick > gpurun_out/r02g/c_${prec}_ab$ab.json 2>/dev/null
    python - <<PY
import json
d=json.load(open('gpurun_out/r02g/c_${prec}_ab$ab.json'))
print('ablate=$ab $prec layers(us) %s' % ([round(1e3*x,1) for x in d['layer_launch_ms']]))
PY
  done
done
