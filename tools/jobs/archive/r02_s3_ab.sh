# A/B of two library builds through FR_LIB: $1 = base .so, rest = "model batch precision" triples (quoted)
cd $GRAFT_REPO_ROOT
O=gpurun_out/s3_ab; mkdir -p $O
BASE=$1; shift
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "bf16_chain or fp8_chain or streaming_push or scores_within or random_streaming or nan_in" 2>&1 | tail -2 | tee $O/parity.txt || exit 1
for rnd in 1 2 3; do
for cfg in "$@"; do
read M B P <<< "$cfg"
for lib in base new; do
if [ $lib = base ]; then export FR_LIB=$GRAFT_REPO_ROOT/$BASE; else unset FR_LIB; fi
timeout -k 10 200 python bench.py --model $M --batch $B --precision $P --quick > $O/line.json 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/line.json').read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('round $rnd $M $B $P $lib: %.2f M inf/s  launch %.2f us (%s)' % (d['value']/1e6, 1e3*r.get('avg_launch_ms',0), r.get('kernel','')[:40]))" | tee -a $O/ab.txt
done; done; done
