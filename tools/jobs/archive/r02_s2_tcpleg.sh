cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --legs value,tcp 2>gpurun_out/tcpleg_err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('value %.1f M' % (d['value']/1e6)); print(d['tcp_streaming']['value']); print(json.dumps(d['tcp_serving_with_replies'], indent=1))"
tail -2 gpurun_out/tcpleg_err.txt
