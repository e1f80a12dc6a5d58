cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ctrace -- python3 $R/bench.py --model C --batch 4096 --precision bf16 --no-cpu-baseline --no-model-c --steps 200 --warmup 40 > $R/gpurun_out/ctrace.log 2>&1
cd $R
f=$(ls -t gpurun_out/ctrace/*/*_kernel_trace.csv | head -1)
python tools/trace_timeline.py $f fc_lp_gemm
