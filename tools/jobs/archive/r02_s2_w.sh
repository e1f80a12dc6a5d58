cd $GRAFT_REPO_ROOT
H=gpu-fpga-recommendation-system_amd/host
O=gpurun_out/s2_reply; mkdir -p $O
run() {  # model batch prec threads total window interval extra
  M=$1; B=$2; P=$3; T=$4; TOTAL=$5; W=$6; IV=$7; shift 7
  PORT=$((20000 + RANDOM % 20000))
  $H/fleetrec_server --model $M --batch $B --precision $P --threads $T --port $PORT --total $TOTAL --tables hash --weights uniform --stream --reply "$@" > $O/srv.txt 2>&1 &
  SP=$!
  sleep 1
  timeout 120 $H/fleetrec_sender --model $M --batch $B --threads $T --port $PORT --indices uniform --reply --window $W --interval-us $IV "$@" > $O/snd.txt 2>&1 &
  NP=$!
  wait $SP; wait $NP 2>/dev/null
  echo "Model-$M batch $B $P $* connections $T window $W interval $IV us: $(grep 'first connection' $O/srv.txt | sed 's/first connection -> last scores: //') | $(grep '^latency' $O/snd.txt | sed 's/latency request sent -> scores received //')"
}
run B 1024 bf16 4 200000 128 0
run B 1024 bf16 8 200000 128 0
run B 1024 bf16 8 200000 128 0 --per-bank
run B 1024 bf16 4 8000 128 500
run B 1024 bf16 4 30000 128 100
