#!/bin/bash
# stage-pipeline roofline legs measured with the workers side by side: HIP events against rocprofv3 averages of the same run
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_conc_check; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for prec in bf16 fp8 f32; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$prec -- python3 $R/bench.py --roofline-only --model C --batch 4096 --precision $prec > $O/$prec.log 2>&1 || echo "failed $prec"
  f=$(ls $O/st_$prec/*/*kernel_stats.csv | head -1); cp $f $O/${prec}_kernel_stats.csv; rm -rf $O/st_$prec
  python3 - <<PY
import json,csv
d=[json.loads(l) for l in open("$O/$prec.log") if l.startswith('{"metric"')][-1]
rf=d["roofline"]; print("$prec", rf["kernel_name"], "events %.1f us" % (1e3*rf["avg_launch_ms"]), "conc", rf.get("concurrent_launches"), "frac %.3f" % rf["frac"])
print("  layers", d.get("layer_kernels"), [round(1e3*x,1) for x in d.get("layer_launch_ms",[])], [round(x,2) for x in d.get("layer_concurrency",[])])
for r in csv.DictReader(open("$O/${prec}_kernel_stats.csv")):
    if "gemm" in r["Name"]: print("  prof %-50s calls %5s avg %.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
