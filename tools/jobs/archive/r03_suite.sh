# round 3: full GPU suite + smoke + the driver's bench command (N = 1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_suite; mkdir -p $O
timeout -k 10 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee $O/pytest_tail.txt
grep -q "failed\|error" $O/pytest_tail.txt && exit 1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SECONDS=0; timeout -k 10 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench wall $SECONDS s rc $?"; tail -c 300 $O/bench_line.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_suite/bench_line.json").read().strip().splitlines()[-1])
print("value %.2f M  roofline %.3f (%s)" % (d["value"] / 1e6, d["roofline"]["frac"], d["roofline"]["kernel_name"]))
for c in d.get("configs", []):
    r = c.get("roofline", {})
    print("%-110s %8.2f M  kernel %s  frac %.3f  %.1f us" % (c.get("workload", "")[:110], (c.get("value") or 0) / 1e6, r.get("kernel_name"), r.get("frac", 0), 1e3 * r.get("avg_launch_ms", 0)))
for k in ("gather", "gather_per_bank"):
    g = d.get(k, {})
    print(k, g.get("kernel_name"), "%.1f us frac %.3f" % (1e3 * g.get("avg_launch_ms", 0), g.get("frac", 0)))
PY
