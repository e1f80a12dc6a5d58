cd $GRAFT_REPO_ROOT
H=gpu-fpga-recommendation-system_amd/host
O=gpurun_out/s2_reply; mkdir -p $O

run() {  # threads total window interval small
  T=$1; TOTAL=$2; W=$3; IV=$4; SB=$5
  PORT=$((20000 + RANDOM % 20000))
  $H/fleetrec_server --model A --batch 256 --threads $T --port $PORT --total $TOTAL --tables hash --weights uniform --stream --reply --small-block $SB > $O/srv.txt 2>&1 &
  SP=$!
  sleep 1
  timeout 120 $H/fleetrec_sender --model A --batch 256 --threads $T --port $PORT --indices uniform --reply --window $W --interval-us $IV > $O/snd.txt 2>&1 &
  NP=$!
  wait $SP; wait $NP 2>/dev/null
  echo "small-block $SB window $W interval $IV us: $(grep 'first connection' $O/srv.txt | sed 's/first connection -> last scores: //') | $(grep '^latency' $O/snd.txt | sed 's/latency request sent -> scores received //')"
}
for SB in 4; do
run 4 400000 256 0 $SB
run 4 150000 16 0 $SB
run 4 60000 4 0 $SB
run 4 30000 1 0 $SB
run 4 40000 256 100 $SB
run 4 12000 256 500 $SB
done 2>&1 | tee $O/sweep_small.txt
