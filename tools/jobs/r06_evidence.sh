#!/bin/bash
# round 6 evidence, part 1 (one box): rocprofv3 --kernel-trace --stats of every roofline leg's own command (single stream) and of the multi-stream
# throughput loops that run faster than their one-stream kernel fraction suggests; per leg the kernel-stats CSV + the kernel-duration
# medians / span per launch from the same run's trace (tools/trace_kernel_median.py).  Part 2: tools/jobs/r06_evidence_pmc.sh; part 3: r06_evidence_line.sh.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_evidence
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
st() { # name, bench args...   (ONLY=<prefix> in the environment: just the legs whose name starts with it)
  n=$1; shift
  case "$n" in ${ONLY:-}*) ;; *) return 0;; esac
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$n -- python3 $R/bench.py "$@" > $O/stats_$n.log 2>&1 || echo "stats $n failed"
  f=$(ls $O/stats_$n/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/${n}_kernel_stats.csv
  t=$(ls $O/stats_$n/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$t" ] && python3 $R/tools/trace_kernel_median.py $t $O/${n}_kernel_durations.json > $O/${n}_kernel_durations.txt 2>&1
  [ -n "$t" ] && python3 -c "
import csv,sys
w=csv.writer(open(sys.argv[2],'w'))
w.writerow(['Queue_Id','Stream_Id','Kernel_Name','Start_Timestamp','End_Timestamp'])
for r in csv.DictReader(open(sys.argv[1])): w.writerow([r['Queue_Id'],r['Stream_Id'],r['Kernel_Name'].split('(')[0],r['Start_Timestamp'],r['End_Timestamp']])" $t $O/${n}_kernel_trace_min.csv   # kept in gpurun_out (scratch) so that the reduction can be re-run here
  rm -rf $O/stats_$n
  echo "stats $n done"
}
st roofline --roofline-only
st value_4streams --legs none --quick
st gather_per_table_uniform --roofline-only --legs gather --gather-law uniform --no-gather-ab
st gather_per_table_zipf --roofline-only --legs gather --gather-law zipf --no-gather-ab
st gather_per_bank_uniform --roofline-only --legs bank --no-gather-ab
for prec in f32 bf16 fp8; do st C4096_$prec --roofline-only --model C --batch 4096 --precision $prec; done
st B1024_bf16 --roofline-only --model B --batch 1024 --precision bf16
st B1024_bf16_per_bank --roofline-only --model B --batch 1024 --precision bf16 --per-bank
st B1024_f32 --roofline-only --model B --batch 1024 --precision f32
st B1024_fp8 --roofline-only --model B --batch 1024 --precision fp8
st A256_bf16 --roofline-only --model A --batch 256 --precision bf16
st A256_fp8 --roofline-only --model A --batch 256 --precision fp8
st A256_bf16_4streams --throughput-only --model A --batch 256 --precision bf16
st C4096_bf16_per_bank --roofline-only --model C --batch 4096 --precision bf16 --per-bank
st C4096_fp8_per_bank --roofline-only --model C --batch 4096 --precision fp8 --per-bank
st C4096_bf16_per_bank_4streams --throughput-only --model C --batch 4096 --precision bf16 --per-bank
st C4096_fp8_per_bank_4streams --throughput-only --model C --batch 4096 --precision fp8 --per-bank
st B1024_bf16_4streams --throughput-only --model B --batch 1024 --precision bf16
st B1024_f32_4streams --throughput-only --model B --batch 1024 --precision f32
ls $O | head -80
