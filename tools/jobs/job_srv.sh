cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_server.py -x -q 2>&1 | tail -5
H=gpu-fpga-recommendation-system_amd/host
$H/fleetrec_server --model A --batch 256 --threads 4 --port 19000 --total 4000 --tables hash --weights uniform --latency > gpurun_out/latency_A.txt 2>&1 &
SP=$!
sleep 1
timeout 120 $H/fleetrec_sender --model A --batch 256 --threads 4 --port 19000 --interval-us 500 > /dev/null 2>&1
wait $SP
tail -6 gpurun_out/latency_A.txt
