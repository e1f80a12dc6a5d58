cd $GRAFT_REPO_ROOT
H=gpu-fpga-recommendation-system_amd/host
O=gpurun_out/s2_reply; mkdir -p $O
run() {  # threads total window
  T=$1; TOTAL=$2; W=$3
  PORT=$((20000 + RANDOM % 20000))
  $H/fleetrec_server --model A --batch 256 --threads $T --port $PORT --total $TOTAL --tables hash --weights uniform --stream --reply > $O/srv.txt 2>&1 &
  SP=$!
  sleep 1
  timeout 120 $H/fleetrec_sender --model A --batch 256 --threads $T --port $PORT --indices uniform --reply --window $W > $O/snd.txt 2>&1 &
  NP=$!
  wait $SP; wait $NP 2>/dev/null
  echo "connections $T window $W: $(grep 'first connection' $O/srv.txt | sed 's/first connection -> last scores: //') | $(grep '^latency' $O/snd.txt | sed 's/latency request sent -> scores received //')"
}
run 4 1000000 256
run 4 1000000 512
run 4 1000000 1024
run 8 1000000 256
