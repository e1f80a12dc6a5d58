# serving with replies: --flush-min sweep now that blocks of <= 8 batches cost n + 4 short launches instead of one 135-us fused launch
cd $GRAFT_REPO_ROOT
H=gpu-fpga-recommendation-system_amd/host
O=gpurun_out/s3_reply2; mkdir -p $O
run() {  # flushmin total window interval
  FM=$1; TOTAL=$2; W=$3; IV=$4
  PORT=$((20000 + RANDOM % 20000))
  $H/fleetrec_server --model A --batch 256 --threads 4 --port $PORT --total $TOTAL --tables hash --weights uniform --stream --reply --flush-min $FM > $O/srv.txt 2>&1 &
  SP=$!
  sleep 1
  timeout 120 $H/fleetrec_sender --model A --batch 256 --threads 4 --port $PORT --indices uniform --reply --window $W --interval-us $IV > $O/snd.txt 2>&1 &
  NP=$!
  wait $SP; wait $NP 2>/dev/null
  echo "flush-min $FM window $W interval $IV us: $(grep 'first connection' $O/srv.txt | sed 's/first connection -> last scores: //') | $(grep '^latency' $O/snd.txt | sed 's/latency request sent -> scores received //')"
}
for rnd in 1 2; do
for FM in 32 16 8 4; do
run $FM 400000 256 0
run $FM 300000 64 0
run $FM 200000 32 0
run $FM 150000 16 0
run $FM 100000 8 0
run $FM 40000 256 50
run $FM 40000 256 100
done; done 2>&1 | tee $O/sweep.txt
