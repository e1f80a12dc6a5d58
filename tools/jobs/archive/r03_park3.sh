# the parked-slice kernel's producers: what a 5 us slot waits for.  Ablations (experiments build, wrong scores): 1 = no row loads, 2 = every row load reads row 0
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_park3; mkdir -p $O
P=$GRAFT_REPO_ROOT/gpu-fpga-recommendation-system_amd
export FR_LIB=$P/libfleetrec_exp.so
for abl in 0 2 1 4 5 6 7; do
echo "=== FR_FUSED_HS_ABLATE=$abl" | tee -a $O/stamps.txt
FR_FUSED_HS_ABLATE=$abl timeout -k 10 200 python tools/experiments/fused_hk_stamps.py 64 B 2>&1 | tail -27 | tee -a $O/stamps.txt
done
