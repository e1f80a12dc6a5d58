# cache-policy bits on the persistent bf16 kernel's WEIGHT loads (hand-built variants of the experiments library, -DFR_HS_W_AUX=n:
# 1 = sc0, 2 = nt, 16 = sc1, 17 = sc0 sc1): does keeping the weight stream out of L1 leave the gathered rows more of it?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_waux; mkdir -p $O
P=$GRAFT_REPO_ROOT/gpu-fpga-recommendation-system_amd
for rep in 1 2; do
for v in "" _w1 _w2 _w16 _w17; do
FR_LIB=$P/libfleetrec_exp$v.so timeout -k 10 300 python bench.py --model B --batch 1024 --precision bf16 2>$O/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('exp$v B 1024 bf16: %.2f M inf/s  launch %.2f us' % (d['value']/1e6, 1e3*r['avg_launch_ms']))" | tee -a $O/ab.txt
done; done
