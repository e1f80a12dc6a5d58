cd $GRAFT_REPO_ROOT
O=gpurun_out/s2_bpre; mkdir -p $O
for r in 108 208 408 112 212 412 -1 -2; do echo "=== knob $r, 16 batches per launch"; FR_FUSED_H_RING=$r timeout 300 python tools/experiments/fused_h_stamps.py 16 2>&1 | grep -v amdgpu.ids | grep -v "^wave [1-35-7]"; done > $O/stamps.txt 2>&1
cat $O/stamps.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "bf16 or streaming or random_custom or per_bank_gather" 2>&1 | tail -4
for r in 208 408 212; do
FR_FUSED_H_RING=$r timeout 300 python bench.py --model B --batch 1024 --precision bf16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B bf16 knob $r', d['value']/1e6, d['roofline']['avg_launch_ms'])"
done
