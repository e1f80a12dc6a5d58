cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bf16_chain or streaming or random_custom or sharded_mode" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
python tools/experiments/fused_h_stamps.py 2>&1 | grep -v amdgpu
for rep in 1 2; do
timeout 600 python bench.py --model B --batch 1024 --precision bf16 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B1024 bf16 value %.1f M  launch %.1f us frac %.3f' % (d['value']/1e6, d['roofline']['avg_launch_ms']*1e3, d['roofline']['frac']))"
done
timeout 600 python bench.py --model A --batch 256 --precision bf16 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('A256 bf16 value %.1f M  launch %.1f us frac %.3f' % (d['value']/1e6, d['roofline']['avg_launch_ms']*1e3, d['roofline']['frac']))"
