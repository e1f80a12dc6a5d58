cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for p in bf16 fp8; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gstat_$p -- python3 $R/bench.py --model C --batch 4096 --precision $p --threads 1 --depth 1 --no-cpu-baseline --no-model-c --steps 200 --warmup 40 > $R/gpurun_out/gstat_$p.log 2>&1
f=$(ls -t $R/gpurun_out/gstat_$p/*/*_kernel_stats.csv | head -1)
echo "== $p"; head -6 $f | cut -c1-150
done
