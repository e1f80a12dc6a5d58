# the driver's bench command and the rocprofv3 kernel stats of the roofline leg on ONE box (the pair profiles/README.md quotes)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pair; mkdir -p $O
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_roofline -- python3 $R/bench.py --roofline-only > $O/stats_roofline.log 2>&1
cp $(ls $O/stats_roofline/*/*kernel_stats.csv | head -1) $O/stats_roofline_kernel_stats.csv
sed -n 2p $O/stats_roofline_kernel_stats.csv | cut -c1-120
python3 -c "
import json; d=json.load(open('$O/bench_line.json')); print('value %.2f M frac %.4f launch %.1f us' % (d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms']*1e3))"
