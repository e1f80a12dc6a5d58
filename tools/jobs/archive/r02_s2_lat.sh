cd $GRAFT_REPO_ROOT
H=gpu-fpga-recommendation-system_amd/host
O=gpurun_out/s2_lat; mkdir -p $O
for z in 0 1; do
PORT=$((20000 + RANDOM % 20000))
FR_SUBMIT_ZEROCOPY=$z $H/fleetrec_server --model A --batch 256 --threads 4 --port $PORT --total 8000 --tables hash --weights uniform --latency > $O/srv_$z.txt 2>&1 &
SP=$!
sleep 1
timeout 120 $H/fleetrec_sender --model A --batch 256 --threads 4 --port $PORT --indices uniform --interval-us 500 > $O/snd.txt 2>&1 &
NP=$!
wait $SP; kill $NP 2>/dev/null; wait $NP 2>/dev/null
echo "FR_SUBMIT_ZEROCOPY=$z"; grep "^latency" $O/srv_$z.txt
done
