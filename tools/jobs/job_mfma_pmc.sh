cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/mfma_pmc
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/mfma_pmc/p1 -- python3 $R/bench.py --roofline-only > $R/gpurun_out/mfma_pmc/p1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/gpurun_out/mfma_pmc/p2 -- python3 $R/bench.py --roofline-only > $R/gpurun_out/mfma_pmc/p2.log 2>&1
cd $R
python3 - <<'PY'
import csv,glob,collections
for p in ('p1','p2'):
    agg=collections.defaultdict(list)
    for f in glob.glob('gpurun_out/mfma_pmc/%s/*/*counter_collection.csv'%p):
        for r in csv.DictReader(open(f)):
            if 'fused_tile' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(p, k, len(v), sum(v)/len(v))
PY
tail -3 gpurun_out/mfma_pmc/p1.log | cut -c1-300
