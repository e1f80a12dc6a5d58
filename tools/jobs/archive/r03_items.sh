# Model-B 1024 bf16: 64 batches (65536 items, 4 tiles per workgroup) vs 128 / 256 batches per launch (8 / 16 tiles): the first tile's exposed prologue amortised
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_items; mkdir -p $O
EXP=$GRAFT_REPO_ROOT/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for rnd in 1 2; do for cfg in "64 65536" "128 131072" "256 262144"; do for pb in "" "--per-bank"; do
read G IT <<< "$cfg"
FR_LIB=$EXP FR_FUSED_ITEMS=$IT timeout -k 10 300 python bench.py --model B --batch 1024 --precision bf16 --group $G $pb > $O/line.json 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/line.json').read().strip().splitlines()[-1]); r=d['roofline']; print('round $rnd group $G items $IT $pb: %.2f M inf/s   one stream: %.1f us per launch (%s)' % (d['value']/1e6, 1e3*r['avg_launch_ms'], r['kernel_name']))" | tee -a $O/ab.txt
done; done; done
