# Model-A 256 bf16 through the persistent kernel (one tile per workgroup): kernel duration vs launch period on one stream, against the chunked kernel
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_hsA; mkdir -p $O
EXP=$GRAFT_REPO_ROOT/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
export FR_LIB=$EXP
cd /tmp && export TMPDIR=/tmp
for hk in 0 1; do
export FR_FUSED_HK=$hk
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$hk -- python3 $GRAFT_REPO_ROOT/bench.py --roofline-only --model A --batch 256 --precision bf16 > $O/trace_$hk.log 2>&1
python3 - $O/trace_$hk <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'fused' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp'])); rows=rows[-100:]
dur=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
per=[(int(rows[i+1]['Start_Timestamp'])-int(rows[i]['Start_Timestamp']))/1e3 for i in range(len(rows)-1)]
per=[p for p in per if p<300]
print(rows[0]['Kernel_Name'][:60], 'scratch', rows[0]['Scratch_Size'], 'grid', rows[0]['Grid_Size_X'], 'dur avg %.1f us'%(sum(dur)/len(dur)), 'period avg %.1f us'%(sum(per)/len(per)))
PY
grep -o '"avg_launch_ms": [0-9.]*' $O/trace_$hk.log
done
