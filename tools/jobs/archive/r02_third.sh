# round 2, third GPU call: gather-variant parity, gather legs with the kernel A/B, PMC passes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "variants or per_bank" 2>&1 | tail -15
timeout 900 python bench.py --quick --legs gather,bank > gpurun_out/r02c/gather_ab.json 2> gpurun_out/r02c/gather_ab.err; tail -c 800 gpurun_out/r02c/gather_ab.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02c/gather_ab.json'))
for k in ('gather','gather_per_bank'):
    g=d.get(k,{})
    print(k, g.get('avg_launch_ms'), g.get('frac'), json.dumps(g.get('kernel_ab')))
    if 'zipf_1.05' in g: print(' zipf', g['zipf_1.05'].get('avg_launch_ms'), g['zipf_1.05'].get('frac'), json.dumps(g['zipf_1.05'].get('kernel_ab')))
PY
timeout 1800 bash tools/pmc_passes.sh > gpurun_out/r02c/pmc.log 2>&1; tail -6 gpurun_out/r02c/pmc.log
