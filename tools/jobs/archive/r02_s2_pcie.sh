cd $GRAFT_REPO_ROOT
O=gpurun_out/s2_pcie; mkdir -p $O
for td in "2 2" "4 1" "4 2" "3 2" "8 1"; do set -- $td
  timeout 300 python bench.py --legs value,pcie --threads $1 --depth $2 > $O/t$1_d$2.json 2>$O/err.txt
  python - $O/t$1_d$2.json "$td" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], 'value %.1f M  streaming %.1f M  per-batch %.1f M' % (d['value']/1e6, d['pcie_inclusive_streaming']['value']/1e6, d['pcie_inclusive']['value']/1e6), flush=True)
PY
done
