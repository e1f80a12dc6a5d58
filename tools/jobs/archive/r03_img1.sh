# the whole-image form of the fp8 persistent kernel (D = -2): parity under the experiments build, then A/B against the chunked fp8 kernel and the ring form
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_img1; mkdir -p $O
EXP=$GRAFT_REPO_ROOT/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
FR_LIB=$EXP FR_FUSED_HK=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fp8_persistent or fp8_chain" 2>&1 | tail -6 | tee $O/parity.txt
grep -q "failed\|error" $O/parity.txt && exit 1
for rnd in 1 2; do
for cfg in "0 1 B 1024 0" "1 1 B 1024 0" "1 0 B 1024 0" "0 1 A 256 0" "1 1 A 256 128" "1 1 A 256 256" "1 0 A 256 256"; do
read HK IMG M B G <<< "$cfg"
GA=""; [ "$G" != "0" ] && GA="--group $G"
FR_LIB=$EXP FR_FUSED_HK=$HK FR_FUSED_HS_IMG=$IMG timeout -k 10 300 python bench.py --model $M --batch $B --precision fp8 $GA > $O/line.json 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/line.json').read().strip().splitlines()[-1]); r=d['roofline']; print('round $rnd hk=$HK img=$IMG $M $B fp8 group $G: %.2f M inf/s   one stream %.1f us per launch (%s)' % (d['value']/1e6, 1e3*r['avg_launch_ms'], r['kernel_name']))" | tee -a $O/ab.txt
done; done
