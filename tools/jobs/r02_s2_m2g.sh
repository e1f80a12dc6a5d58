cd $GRAFT_REPO_ROOT
O=gpurun_out/s2_m2g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "streaming or driver_loop or host_fed or scores_within or known" 2>&1 | tail -3
for i in 1 2 3; do
for lib in tools/experiments/variants/libfleetrec_before_m2gather.so gpu-fpga-recommendation-system_amd/libfleetrec.so; do
  FR_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --legs value,roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib value %.2f M  launch %.1f us  frac %.4f' % (d['value']/1e6, d['roofline']['avg_launch_ms']*1e3, d['roofline']['frac']), flush=True)"
done; done | tee $O/ab.txt
