cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/cfg
cd /tmp && export TMPDIR=/tmp
run() { # name, args...
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cfg/$name -- python3 $R/bench.py "$@" --no-cpu-baseline --no-model-c --steps 1000 --warmup 100 > $R/gpurun_out/cfg/$name.log 2>&1
  f=$(ls -t $R/gpurun_out/cfg/$name/*/*_kernel_stats.csv | head -1)
  cp $f $R/gpurun_out/cfg/$name.kernel_stats.csv
  tail -1 $R/gpurun_out/cfg/$name.log | cut -c1-330 > $R/gpurun_out/cfg/$name.bench_line.json
  echo "$name: $(python3 -c "import json;d=json.load(open('$R/gpurun_out/cfg/$name.bench_line.json'.replace('.json','.json')));print(round(d['value']/1e6,2),'M inf/s',d['config'].get('fc_tflops'))" 2>/dev/null)"
}
run B1024_f32 --model B --batch 1024 --precision f32
run B1024_bf16 --model B --batch 1024 --precision bf16
run B1024_fp8 --model B --batch 1024 --precision fp8
run C4096_f32 --model C --batch 4096 --precision f32
run C4096_bf16 --model C --batch 4096 --precision bf16
run C4096_fp8 --model C --batch 4096 --precision fp8
run A256_bf16 --model A --batch 256 --precision bf16
