# where the persistent bf16 kernel starts to pay: 1.5 tiles per CU (Model-B: 24 batches of 1024 per launch; Model-A: 96 of 256), chunked vs persistent forced
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_thr; mkdir -p $O
EXP=$GRAFT_REPO_ROOT/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for rnd in 1 2; do for cfg in "B 1024 24" "B 1024 20" "A 256 96" "A 256 80"; do for hk in 0 1; do
read M B G <<< "$cfg"
FR_LIB=$EXP FR_FUSED_HK=$hk timeout -k 10 300 python bench.py --model $M --batch $B --precision bf16 --group $G > $O/line.json 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/line.json').read().strip().splitlines()[-1]); r=d['roofline']; print('round $rnd $M $B group $G hk=$hk: %.2f M inf/s   one stream: %.1f us per launch (%s)' % (d['value']/1e6, 1e3*r['avg_launch_ms'], r['kernel_name']))" | tee -a $O/ab.txt
done; done; done
