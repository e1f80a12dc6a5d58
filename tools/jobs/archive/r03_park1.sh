# the persistent bf16 kernel with the gather spread over FC1 AND FC2 (next tile's slices 2, 3 parked in consumed R1 rows): parity subset, then a
# same-job A/B against the previous build (libfleetrec_prev.so = the kernel whose gather only runs under FC1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_park1; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "bf16 or persistent_fused or dense_block or streaming_push or groups_above_64 or fixtures" 2>&1 | tail -6 | tee $O/pytest_tail.txt
grep -q "failed\|error" $O/pytest_tail.txt && exit 1
P=$GRAFT_REPO_ROOT/gpu-fpga-recommendation-system_amd
for rep in 1 2; do
for lib in libfleetrec_prev.so libfleetrec.so; do
for cfg in "B 1024 0" "B 1024 1" "A 256 0"; do
read M B PB <<< "$cfg"
F=""; [ $PB = 1 ] && F="--per-bank"
FR_LIB=$P/$lib timeout -k 10 300 python bench.py --model $M --batch $B --precision bf16 $F 2>$O/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$lib $M $B per_bank=$PB: %.2f M inf/s  launch %.2f us frac %.3f (%s)' % (d['value']/1e6, 1e3*r['avg_launch_ms'], r['frac'], r.get('kernel_name','')[:50]))" | tee -a $O/ab.txt
done; done; done
FR_LIB=$P/libfleetrec_exp.so timeout -k 10 200 python tools/experiments/fused_hk_stamps.py 64 B 2>&1 | tail -40 | tee $O/stamps.txt
