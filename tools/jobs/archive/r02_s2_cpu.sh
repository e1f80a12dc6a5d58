cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --legs value,roofline,cpu 2>gpurun_out/cpu_err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['cpu_baseline']; print(json.dumps({k:v for k,v in c.items() if k not in ('sample','gather_thread_probe_s')}, indent=1))"
tail -3 gpurun_out/cpu_err.txt
