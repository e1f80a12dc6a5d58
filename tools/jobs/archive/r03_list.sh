# fr_worker_push_device_list: its test, then the one-stream roofline legs fed by native list pushes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_list; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_abi.py -m gpu -q -x -k "push_device_list or streaming_push or abi" 2>&1 | tail -4 | tee $O/pytest_tail.txt
grep -q "failed\|error" $O/pytest_tail.txt && exit 1
for cfg in "A 256 f32 0" "A 256 bf16 256" "A 256 bf16 64" "A 256 fp8 64" "B 1024 bf16 0" "B 1024 fp8 0"; do
read M B P G <<< "$cfg"
GA=""; [ "$G" != "0" ] && GA="--group $G"
timeout -k 10 300 python bench.py --roofline-only --model $M --batch $B --precision $P $GA 2>$O/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$M $B $P group $G: one stream %.2f us per launch, frac %.3f (%s)' % (1e3*r['avg_launch_ms'], r['frac'], r['kernel_name']))" | tee -a $O/out.txt
done
