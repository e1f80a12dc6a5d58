# kernel durations of the Model-C chain with ONE worker on ONE stream (no overlap between kernels): what each launch costs alone
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s3_cchain; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for prec in fp8 bf16; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$prec -- python3 $R/bench.py --model C --batch 4096 --precision $prec --quick --threads 1 --depth 1 > $O/line_$prec.json 2> $O/err_$prec.txt
cp $(ls $O/st_$prec/*/*kernel_stats.csv | head -1) $O/C4096_${prec}_one_stream_kernel_stats.csv
echo "== $prec"; grep -v "fill_\|pack_weights\|rocclr\|stats_kernel" $O/C4096_${prec}_one_stream_kernel_stats.csv | sed 's/(HIP_vector[^"]*"/"/; s/(FrPipeArgs)//' | cut -c1-160
python3 -c "
import json; d=json.loads(open('$O/line_$prec.json').read().strip().splitlines()[-1]); print('value %.2f M inf/s  ms/step %.4f' % (d['value']/1e6, d['ms_per_step']))"
done
