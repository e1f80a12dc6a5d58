cd $GRAFT_REPO_ROOT
H=gpu-fpga-recommendation-system_amd/host
O=gpurun_out/s2_reply; mkdir -p $O
run() {  # threads total window interval flush_blocks
  T=$1; TOTAL=$2; W=$3; IV=$4; FB=$5
  PORT=$((20000 + RANDOM % 20000))
  $H/fleetrec_server --model A --batch 256 --threads $T --port $PORT --total $TOTAL --tables hash --weights uniform --stream --reply --flush-min $FB > $O/srv.txt 2>&1 &
  SP=$!
  sleep 1
  timeout 120 $H/fleetrec_sender --model A --batch 256 --threads $T --port $PORT --indices uniform --reply --window $W --interval-us $IV > $O/snd.txt 2>&1 &
  NP=$!
  wait $SP; wait $NP 2>/dev/null
  echo "flush-min $FB window $W interval $IV us: $(grep 'first connection' $O/srv.txt | sed 's/first connection -> last scores: //') | $(grep '^latency' $O/snd.txt | sed 's/latency request sent -> scores received //')"
}
for FB in 1 32 16 9999; do
run 4 400000 256 0 $FB
run 4 300000 64 0 $FB
run 4 150000 16 0 $FB
run 4 60000 4 0 $FB
run 4 40000 256 100 $FB
run 4 12000 256 500 $FB
done 2>&1 | tee $O/sweep.txt
