cd $GRAFT_REPO_ROOT
H=gpu-fpga-recommendation-system_amd/host
O=gpurun_out/s2_tcpB; mkdir -p $O
run() {  # model batch threads total extra...
  M=$1; B=$2; T=$3; TOTAL=$4; shift 4
  PORT=$((20000 + RANDOM % 20000))
  $H/fleetrec_server --model $M --batch $B --threads $T --port $PORT --total $TOTAL --tables hash --weights uniform "$@" > $O/srv.txt 2>&1 &
  SP=$!
  sleep 1
  EXTRA=""; for a in "$@"; do [ "$a" = "--per-bank" ] && EXTRA="--per-bank"; done
  timeout 120 $H/fleetrec_sender --model $M --batch $B --threads $T --port $PORT --indices uniform $EXTRA > $O/snd.txt 2>&1 &
  NP=$!
  wait $SP
  kill $NP 2>/dev/null; wait $NP 2>/dev/null
  echo "model $M batch $B threads $T $*: $(grep "first connection" $O/srv.txt)"
}
run B 1024 4 200000 --stream --precision bf16
run B 1024 8 200000 --stream --precision bf16
run B 1024 8 200000 --stream --precision bf16 --per-bank
run A 256 4 1000000 --stream
