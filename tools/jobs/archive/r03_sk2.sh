# split-K direct GEMM vs the LDS-tiled kernels on ONE stream (kernel durations without co-resident kernels), Model-C 4096 bf16 and fp8
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_sk2; mkdir -p $O
export FR_LIB=$GRAFT_REPO_ROOT/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
cd /tmp && export TMPDIR=/tmp
for P in bf16 fp8; do
for sk in 0 1; do
export FR_LP_GEMM_SPLITK=$sk
echo "=== $P FR_LP_GEMM_SPLITK=$sk, one driver thread, one worker"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_${P}_$sk -- python3 $GRAFT_REPO_ROOT/bench.py --model C --batch 4096 --precision $P --quick --threads 1 --depth 1 > $O/trace_${P}_$sk.log 2>&1
grep -o '"value": [0-9.]*' $O/trace_${P}_$sk.log | head -1
python3 - $O/trace_${P}_$sk <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage'])>0.5 and 'fill_' not in r['Name']: print('%-70s calls %6s avg %8.2f us  %5s %%' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
done; done 2>&1 | tee $O/kernels.txt
