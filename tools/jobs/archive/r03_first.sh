# round 3, first call: GPU suite on the round-2 head + baselines of the kernels this round works on (this box's numbers)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_first; mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest_tail.txt
for cfg in "B 1024 bf16" "A 256 bf16" "B 1024 fp8" "C 4096 bf16" "C 4096 fp8"; do
read M B P <<< "$cfg"
timeout -k 10 300 python bench.py --model $M --batch $B --precision $P --quick > $O/line_${M}_${P}.json 2> $O/err.txt || { tail -3 $O/err.txt; }
python3 -c "
import json; d=json.loads(open('$O/line_${M}_${P}.json').read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$M $B $P: %.2f M inf/s  launch %.2f us (%s) layers %s' % (d['value']/1e6, 1e3*r.get('avg_launch_ms',0), r.get('kernel','')[:40], d.get('layer_launch_ms')))" | tee -a $O/base.txt
done
timeout -k 10 200 python tools/experiments/fused_h_stamps.py 16 2>&1 | tail -30 | tee $O/stamps.txt
