cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s2_timeline; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for prec in fp8 bf16; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$prec -- python3 $R/bench.py --model C --batch 4096 --precision $prec --quick > $O/tr_$prec.log 2>&1
f=$(ls $O/tr_$prec/*/*kernel_trace.csv | head -1)
echo "== C $prec"; python3 $R/tools/trace_timeline.py $f gemm
done
