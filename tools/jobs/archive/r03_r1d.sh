# FC1 weight ring of the persistent bf16 kernel, 6 (shipped) vs 4 fragments, in the clean experiments build (kernel = the product's)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_r1d; mkdir -p $O
EXP=$GRAFT_REPO_ROOT/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for rnd in 1 2; do for r in 6 4; do for pb in "" "--per-bank"; do
FR_LIB=$EXP FR_FUSED_R1D=$r timeout -k 10 300 python bench.py --model B --batch 1024 --precision bf16 $pb 2>$O/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('round $rnd R1D=$r $pb: %.2f M inf/s  launch %.2f us (%s)' % (d['value']/1e6, 1e3*r['avg_launch_ms'], r['kernel_name']))" | tee -a $O/ab.txt
done; done; done
