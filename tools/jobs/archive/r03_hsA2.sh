# Model-A 256 bf16 at the default group of 64 (one tile per CU): chunked kernel vs the persistent kernel forced (clean experiments build)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_hsA2; mkdir -p $O
EXP=$GRAFT_REPO_ROOT/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for rnd in 1 2; do for hk in 0 1; do for g in 64 96; do
FR_LIB=$EXP FR_FUSED_HK=$hk timeout -k 10 300 python bench.py --model A --batch 256 --precision bf16 --group $g > $O/line.json 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/line.json').read().strip().splitlines()[-1]); r=d['roofline']; print('round $rnd hk=$hk group $g: %.2f M inf/s   one stream: %.1f us per launch (%s)' % (d['value']/1e6, 1e3*r['avg_launch_ms'], r['kernel_name']))" | tee -a $O/ab.txt
done; done; done
