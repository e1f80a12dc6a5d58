cd $GRAFT_REPO_ROOT
O=gpurun_out/s2_cstreams; mkdir -p $O
for prec in bf16 fp8; do for td in "2 2" "1 2" "1 4" "2 4" "4 2" "1 8"; do set -- $td
  timeout 300 python bench.py --model C --batch 4096 --precision $prec --threads $1 --depth $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C $prec threads $1 depth $2: %.1f M inf/s' % (d['value']/1e6), flush=True)"
done; done | tee $O/out.txt
