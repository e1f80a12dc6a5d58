# Model-C batch 4096 chain: the gather | out launch as fr_gather_out_kernel (56 VGPRs: fits beside a resident FC1 GEMM workgroup) vs fr_pipeline_kernel<-1>
cd $GRAFT_REPO_ROOT
O=gpurun_out/s3_go; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "tiled_gemm or per_bank_gather_bit_exact" 2>&1 | tail -2 | tee $O/parity.txt || exit 1
for rnd in 1 2; do
for prec in bf16 fp8 f32; do
for im in "" "--per-bank"; do
for V in 0 1; do
FR_GATHER_OUT_KERNEL=$V timeout -k 10 200 python bench.py --model C --batch 4096 --precision $prec --quick $im > $O/line.json 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/line.json').read().strip().splitlines()[-1]); print('round $rnd $prec [$im] light=$V: %.2f M inf/s  %.1f us/step' % (d['value']/1e6, 1e3*d['ms_per_step']))" | tee -a $O/go.txt
done; done; done; done
