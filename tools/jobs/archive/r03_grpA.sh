# Model-A 256 bf16: launch group 64 (chunked kernel, one tile per CU) vs 128 / 192 / 256 (persistent kernel), product library, >= 2 s steady state
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_grpA; mkdir -p $O
for rnd in 1 2; do for g in 64 128 192 256; do
timeout -k 10 300 python bench.py --model A --batch 256 --precision bf16 --group $g > $O/line.json 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/line.json').read().strip().splitlines()[-1]); r=d['roofline']; print('round $rnd group $g: %.2f M inf/s   one stream: %.1f us per launch, frac %.3f (%s)' % (d['value']/1e6, 1e3*r['avg_launch_ms'], r['frac'], r['kernel_name']))" | tee -a $O/ab.txt
done; done
