# launch-group table with the fused streaming path (default) and with the stage pipeline for every push (FR_FUSED=0)
cd $GRAFT_REPO_ROOT
O=gpurun_out/s3_grp; mkdir -p $O
for F in 1 0; do
FR_FUSED=$F timeout -k 10 300 python bench.py --quick --legs groups > $O/line_$F.json 2> $O/err_$F.txt || { tail -3 $O/err_$F.txt; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/line_$F.json').read().strip().splitlines()[-1])
print('FR_FUSED=$F value %.2f M' % (d['value']/1e6))
for r in d['launch_group_table']['rows']: print('  group %2d: %.1f M inf/s, launch %.1f us, push->scores p50 %.1f us' % (r['group'], r['inferences_per_s']/1e6, 1e3*r['launch_ms_one_stream'], 1e3*r['push_to_scores_ms_p50']))" | tee -a $O/grp.txt
done
