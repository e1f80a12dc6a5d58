cd $GRAFT_REPO_ROOT
for prec in fp8 f32; do for st in 2 3 2 3; do
FR_LP_GEMM_STAGES_BIG=$st timeout 600 python bench.py --model C --batch 4096 --precision $prec 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('big_stages=$st $prec value %.2f M  layers(us) %s' % (d['value']/1e6, [round(1e3*x,1) for x in d['layer_launch_ms']]))"
done; done
