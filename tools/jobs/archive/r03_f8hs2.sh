# the fp8 persistent kernel with 65536 items per launch (4 tiles per workgroup instead of 1) against the chunked fp8 kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_f8hs2; mkdir -p $O
EXP=$GRAFT_REPO_ROOT/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for rnd in 1 2; do
for cfg in "0 16384" "1 16384" "1 65536" "0 65536"; do
read HK IT <<< "$cfg"
FR_LIB=$EXP FR_FUSED_HK=$HK FR_FUSED_ITEMS=$IT timeout -k 10 300 python bench.py --model B --batch 1024 --precision fp8 --quick > $O/line.json 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/line.json').read().strip().splitlines()[-1]); r=d['roofline']; print('round $rnd hk=$HK items $IT B 1024 fp8: %.2f M inf/s (%s)' % (d['value']/1e6, r['kernel_name']))" | tee -a $O/ab.txt
done; done
