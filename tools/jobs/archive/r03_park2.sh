# stamps of the parked-slice kernel (producer-side stamps only: the consumers' code is the product's)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_park2; mkdir -p $O
P=$GRAFT_REPO_ROOT/gpu-fpga-recommendation-system_amd
FR_LIB=$P/libfleetrec_exp.so timeout -k 10 200 python tools/experiments/fused_hk_stamps.py 64 B 2>&1 | tail -40 | tee $O/stamps.txt
FR_LIB=$P/libfleetrec_exp.so timeout -k 10 300 python bench.py --model B --batch 1024 --precision bf16 --quick 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('exp lib: %.2f M inf/s, %.1f us per launch of 64 batches' % (d['value']/1e6, 1e3*d['roofline']['avg_launch_ms']))" | tee -a $O/stamps.txt
