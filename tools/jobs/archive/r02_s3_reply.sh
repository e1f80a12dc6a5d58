# serving with replies at mid loads: small host blocks through n + 4 pipelined stage launches (vs 5 n: FR_SMALL_BLOCK_SERIAL=1); --small-block 4 vs 8
cd $GRAFT_REPO_ROOT
H=gpu-fpga-recommendation-system_amd/host
O=gpurun_out/s3_reply; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_server.py tests/test_gpu_parity.py -q -x -k "server or host_fed or latency" 2>&1 | tail -2 | tee $O/parity.txt || exit 1
run() {  # tag serial smallblock total window interval
  TAG=$1; SER=$2; SB=$3; TOTAL=$4; W=$5; IV=$6
  PORT=$((20000 + RANDOM % 20000))
  FR_SMALL_BLOCK_SERIAL=$SER $H/fleetrec_server --model A --batch 256 --threads 4 --port $PORT --total $TOTAL --tables hash --weights uniform --stream --reply --small-block $SB > $O/srv.txt 2>&1 &
  SP=$!
  sleep 1
  timeout 120 $H/fleetrec_sender --model A --batch 256 --threads 4 --port $PORT --indices uniform --reply --window $W --interval-us $IV > $O/snd.txt 2>&1 &
  NP=$!
  wait $SP; wait $NP 2>/dev/null
  echo "$TAG small-block $SB window $W interval $IV us: $(grep 'first connection' $O/srv.txt | sed 's/first connection -> last scores: //') | $(grep '^latency' $O/snd.txt | sed 's/latency request sent -> scores received //')"
}
for rnd in 1 2; do
for cfg in "serial 1 4" "pipelined 0 4" "pipelined 0 8"; do
read TAG SER SB <<< "$cfg"
run $TAG $SER $SB 300000 64 0
run $TAG $SER $SB 150000 16 0
run $TAG $SER $SB 100000 8 0
run $TAG $SER $SB 60000 4 0
run $TAG $SER $SB 40000 256 100
run $TAG $SER $SB 20000 256 200
run $TAG $SER $SB 12000 256 500
done; done 2>&1 | tee $O/sweep.txt
