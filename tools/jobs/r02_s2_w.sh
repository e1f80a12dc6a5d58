cd $GRAFT_REPO_ROOT
H=gpu-fpga-recommendation-system_amd/host
O=gpurun_out/s2_reply; mkdir -p $O
run() {  # threads total window interval
  T=$1; TOTAL=$2; W=$3; IV=$4
  PORT=$((20000 + RANDOM % 20000))
  $H/fleetrec_server --model A --batch 256 --threads $T --port $PORT --total $TOTAL --tables hash --weights uniform --reply > $O/srv.txt 2>&1 &
  SP=$!
  sleep 1
  timeout 120 $H/fleetrec_sender --model A --batch 256 --threads $T --port $PORT --indices uniform --reply --window $W --interval-us $IV > $O/snd.txt 2>&1 &
  NP=$!
  wait $SP; wait $NP 2>/dev/null
  echo "per-batch submit + sync + reply, connections $T window $W interval $IV us: $(grep 'first connection' $O/srv.txt | sed 's/first connection -> last scores: //') | $(grep '^latency' $O/snd.txt | sed 's/latency request sent -> scores received //')"
}
run 4 12000 16 500
run 4 40000 16 100
run 4 100000 4 0
run 4 200000 16 0
run 8 200000 16 0
