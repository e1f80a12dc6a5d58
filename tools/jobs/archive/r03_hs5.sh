cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_hs5; mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $O/pytest_tail.txt
grep -q "failed\|error" $O/pytest_tail.txt && exit 1
for cfg in "B 1024 bf16" "A 256 bf16"; do
read M B P <<< "$cfg"
timeout -k 10 300 python bench.py --model $M --batch $B --precision $P > $O/line_${M}.json 2> $O/err.txt || { tail -3 $O/err.txt; }
python3 -c "
import json; d=json.loads(open('$O/line_${M}.json').read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$M $B $P: %.2f M inf/s  launch %.2f us (%s)' % (d['value']/1e6, 1e3*r.get('avg_launch_ms',0), r.get('kernel','')[:60]))" | tee -a $O/base.txt
done
