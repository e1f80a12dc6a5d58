# Model-A 256 bf16 (group 256) / fp8 (group 64): the kernel's own duration (rocprofv3) beside the one-stream HIP-event figure, whose launches are fed by Python push calls
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_A_prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "bf16 256" "bf16 64" "fp8 64"; do
read P G <<< "$cfg"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_${P}_$G -- python3 $R/bench.py --roofline-only --model A --batch 256 --precision $P --group $G > $O/st_${P}_$G.log 2>&1
python3 - $O/st_${P}_$G $O/st_${P}_$G.log <<'PY'
import csv,glob,sys,json
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
d=[json.loads(l) for l in open(sys.argv[2]) if l.startswith('{') and '"metric"' in l][-1]
r=d['roofline']
for row in csv.DictReader(open(f)):
    if 'fused' in row['Name']: print('%-60s calls %5s avg %8.2f us | live one-stream %.2f us frac %.3f' % (row['Name'][:60], row['Calls'], float(row['AverageNs'])/1e3, 1e3*r['avg_launch_ms'], r['frac']))
PY
done 2>&1 | tee $O/out.txt
