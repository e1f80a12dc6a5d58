cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_hs3; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -x -k "bf16_chain or persistent_fused or dense_block or streaming_push or nan_in or random_custom" 2>&1 | tail -5 | tee $O/parity.txt
grep -q "failed\|error" $O/parity.txt && exit 1
EXP=$GRAFT_REPO_ROOT/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
timeout -k 10 200 python tools/experiments/fused_hk_stamps.py 64 B 2>&1 | tail -40 | tee -a $O/stamps.txt
FR_LIB=$EXP FR_FUSED_ITEMS=16384 timeout -k 10 200 python tools/experiments/fused_hk_stamps.py 16 B 2>&1 | tail -20 | tee -a $O/stamps.txt
run() { # label, env...
  lbl=$1; shift
  for cfg in "B 1024 bf16" "A 256 bf16"; do
    read M B P <<< "$cfg"
    env "$@" timeout -k 10 300 python bench.py --model $M --batch $B --precision $P --quick > $O/line.json 2> $O/err.txt || { tail -3 $O/err.txt; return 1; }
    python3 -c "
import json; d=json.loads(open('$O/line.json').read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$lbl $M $B $P: %.2f M inf/s  launch %.2f us' % (d['value']/1e6, 1e3*r.get('avg_launch_ms',0)))" | tee -a $O/ab.txt
  done
}
for rnd in 1 2; do
run "chunked      " FR_LIB=$EXP FR_FUSED_HK=0 FR_FUSED_ITEMS=16384 || exit 1
run "spec. 16k    " FR_LIB=$EXP FR_FUSED_ITEMS=16384 || exit 1
run "spec. product" A=1 || exit 1
done
