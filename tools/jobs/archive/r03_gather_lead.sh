# round 3: one index load per run of lanes sharing an index column (gather_pack_stream_kernel<..., LEAD>): parity, then A/B in one process per setting
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_gather_lead; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -x -k "gather" 2>&1 | tail -4 | tee $O/parity.txt
grep -q "failed\|error" $O/parity.txt && exit 1
EXP=$GRAFT_REPO_ROOT/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for rnd in 1 2 3; do for lead in 0 1; do
FR_LIB=$EXP FR_GATHER_LEAD=$lead timeout -k 10 300 python bench.py --roofline-only --legs gather,bank --no-gather-ab > $O/line.json 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/line.json').read().strip().splitlines()[-1]); g=d['gather']; b=d['gather_per_bank']; z=g['zipf_1.05']
print('round $rnd lead=$lead: per-table %.2f us (%.3f)  zipf %.2f us  per-bank %.2f us (%.3f)  %s' % (1e3*g['avg_launch_ms'], g['frac'], 1e3*z['avg_launch_ms'], 1e3*b['avg_launch_ms'], b['frac'], b['kernel_name']))" | tee -a $O/ab.txt
done; done
