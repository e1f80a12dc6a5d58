cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --legs value,pcie,tcp 2>gpurun_out/tcpleg_err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('value %.1f M' % (d['value']/1e6)); print('pcie streaming %.1f M' % (d['value_pcie_inclusive']/1e6)); print(d['tcp_streaming'])"
tail -2 gpurun_out/tcpleg_err.txt
