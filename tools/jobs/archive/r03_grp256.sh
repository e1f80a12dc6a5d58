# groups above 64: parity, then Model-A 256 bf16 at group 64 (chunked kernel) vs 128 / 256 (persistent kernel, 2 / 4 tiles per workgroup)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_grp256; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -x -k "groups_above_64 or persistent_fused or streaming_push or bf16_chain or dense_block or driver_loop" 2>&1 | tail -5 | tee $O/parity.txt
grep -q "failed\|error" $O/parity.txt && exit 1
for rnd in 1 2; do for g in 64 128 256; do
timeout -k 10 300 python bench.py --model A --batch 256 --precision bf16 --group $g --quick > $O/line.json 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/line.json').read().strip().splitlines()[-1]); r=d['roofline']; print('round $rnd group $g: %.2f M inf/s   one stream: %.1f us per launch (%s)' % (d['value']/1e6, 1e3*r['avg_launch_ms'], r['kernel_name']))" | tee -a $O/ab.txt
done; done
