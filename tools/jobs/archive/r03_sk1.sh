# split-K direct GEMM for the narrow layers of Model-C at batch 4096 (FC2 2048 -> 512, FC3 512 -> 256): parity, then A/B through the experiments build
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_sk1; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tiled_gemm_model_c or bf16_chain or fp8_chain" 2>&1 | tail -6 | tee $O/pytest_tail.txt
grep -q "failed\|error" $O/pytest_tail.txt && exit 1
export FR_LIB=$GRAFT_REPO_ROOT/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for rep in 1 2; do
for sk in 0 1; do
for cfg in "bf16 0" "bf16 1" "fp8 0" "fp8 1"; do
read P PB <<< "$cfg"
F=""; [ $PB = 1 ] && F="--per-bank"
FR_LP_GEMM_SPLITK=$sk timeout -k 10 300 python bench.py --model C --batch 4096 --precision $P $F 2>$O/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('splitk=$sk C 4096 $P per_bank=$PB: %.2f M inf/s' % (d['value']/1e6))" | tee -a $O/ab.txt
done; done; done
cd /tmp && export TMPDIR=/tmp
for sk in 0 1; do
export FR_LP_GEMM_SPLITK=$sk
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$sk -- python3 $GRAFT_REPO_ROOT/bench.py --model C --batch 4096 --precision bf16 --quick > $O/trace_$sk.log 2>&1
python3 - $O/trace_$sk <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage'])>0.5: print('%-70s calls %6s avg %8.2f us  %5s %%' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
done 2>&1 | tee $O/kernels.txt
