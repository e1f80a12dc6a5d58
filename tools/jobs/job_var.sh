cd $GRAFT_REPO_ROOT
for v in "$@"; do
  cp tools/experiments/variants/$v.so gpu-fpga-recommendation-system_amd/libfleetrec.so
  echo "== $v"
  FR_FUSED_WPE=2 timeout 120 python tools/experiments/fused_stamps.py 32 5000 | grep -E "span|per-wave|end |clock"
done
