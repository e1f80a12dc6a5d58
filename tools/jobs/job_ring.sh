cd $GRAFT_REPO_ROOT
for v in m2_ra12_rb8 m2_ra16_rb8 m2_ra12_rb12 m2_ra8_rb8 m2_ra16_rb12 m2_ra12_rb8; do
  cp tools/experiments/variants/$v.so gpu-fpga-recommendation-system_amd/libfleetrec.so
  timeout 300 python bench.py --no-cpu-baseline --no-model-c --steps 20000 --warmup 2000 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', round(d['value']/1e6,2), round(r['achieved'],1), r['avg_launch_ms'])"
done
