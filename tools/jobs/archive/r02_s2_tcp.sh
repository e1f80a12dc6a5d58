cd $GRAFT_REPO_ROOT
H=gpu-fpga-recommendation-system_amd/host
O=gpurun_out/s2_tcp; mkdir -p $O
run() {  # threads total extra...
  T=$1; TOTAL=$2; shift 2
  PORT=$((20000 + RANDOM % 20000))
  $H/fleetrec_server --model A --batch 256 --threads $T --port $PORT --total $TOTAL --tables hash --weights uniform "$@" > $O/srv.txt 2>&1 &
  SP=$!
  sleep 1
  timeout 120 $H/fleetrec_sender --model A --batch 256 --threads $T --port $PORT --indices uniform > $O/snd.txt 2>&1 &
  NP=$!
  wait $SP
  kill $NP 2>/dev/null; wait $NP 2>/dev/null
  echo "threads $T $*: $(grep "first connection" $O/srv.txt)"
}
run 4 1000000 --stream
run 8 1000000 --stream
run 16 1000000 --stream
run 4 100000
run 8 200000
