cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tiled_gemm or sharded or bf16_chain or fp8_chain" 2>&1 | grep -E "passed|failed|Error" | tail -3
for prec in f32 bf16 fp8; do for st in 2 4; do
FR_LP_GEMM_STAGES=$st timeout 600 python bench.py --model C --batch 4096 --precision $prec 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('stages=$st $prec value %.2f M  layers(us) %s' % (d['value']/1e6, [round(1e3*x,1) for x in d['layer_launch_ms']]))"
done; done
