# round 3: the fp8 form of the persistent wave-specialised kernel: parity, then Model-B 1024 fp8 / Model-A 256 fp8 (group 128) against the chunked fp8 kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_f8hs; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -x -k "fp8 or persistent_fused or groups_above_64" 2>&1 | tail -8 | tee $O/parity.txt
grep -q "failed\|error" $O/parity.txt && exit 1
EXP=$GRAFT_REPO_ROOT/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for rnd in 1 2; do
for cfg in "0 B 1024 0" "1 B 1024 0" "0 A 256 128" "1 A 256 128" "-1 A 256 0"; do
read HK M B G <<< "$cfg"
GA=""; [ "$G" != "0" ] && GA="--group $G"
FR_LIB=$EXP FR_FUSED_HK=$HK timeout -k 10 300 python bench.py --model $M --batch $B --precision fp8 --quick $GA > $O/line.json 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/line.json').read().strip().splitlines()[-1]); r=d['roofline']; print('round $rnd hk=$HK $M $B fp8 group $G: %.2f M inf/s   one stream %.1f us per launch (%s)' % (d['value']/1e6, 1e3*r['avg_launch_ms'], r['kernel_name']))" | tee -a $O/ab.txt
done; done
