# the persistent bf16 kernel with and without the producers' row loads (timing ablation, wrong scores): the consumers' own pace
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_abl; mkdir -p $O
export FR_LIB=$GRAFT_REPO_ROOT/gpu-fpga-recommendation-system_amd/libfleetrec_diag.so
for abl in 0 1; do
echo "=== FR_FUSED_HS_ABLATE=$abl" | tee -a $O/stamps.txt
FR_FUSED_HS_ABLATE=$abl timeout -k 10 200 python tools/experiments/fused_hk_stamps.py 64 B 2>&1 | tail -36 | tee -a $O/stamps.txt
FR_FUSED_HS_ABLATE=$abl timeout -k 10 300 python bench.py --model B --batch 1024 --precision bf16 --quick 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ablate=$abl: %.2f M inf/s, one stream %.1f us per launch of 64 batches' % (d['value']/1e6, 1e3*d['roofline']['avg_launch_ms']))" | tee -a $O/stamps.txt
done
