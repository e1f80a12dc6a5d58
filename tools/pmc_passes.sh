#!/bin/bash
# HBM/fabric traffic, L2 hit rate and MFMA activity of the roofline kernels from PMC counters, as MI355X_MICROARCH.md ("HBM",
# "rocprofv3 PMC slots") prescribes: SEPARATE --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), --kernel-trace only, the
# program itself directly after `--`.  One bench leg per key, so that every kernel name carries ONE workload:
#   fused_m2_A256                    bench.py --legs roofline                      fr_fused_tile_m2_kernel
#   gather_C4096_per_table_uniform   bench.py --legs gather --gather-law uniform   gather_pack_xcd_kernel
#   gather_C4096_per_table_zipf      bench.py --legs gather --gather-law zipf      gather_pack_xcd_kernel
#   gather_C4096_per_bank_uniform    bench.py --legs bank                          gather_pack_xcd_kernel
#   fused_h_B1024_bf16               bench.py --roofline-only --model B --batch 1024 --precision bf16    fr_fused_tile_hs_kernel (round 3; fr_fused_tile_h_kernel before)
#   gemm_C4096_{f32,bf16,fp8}        bench.py --roofline-only --model C --batch 4096 --precision P       the FC1 GEMM kernel of that precision (bf16 / fp8: the
#                                    256 x 256-tile kernel the four workers run side by side; counter passes serialise the launches, so
#                                    bytes per launch are the figure to read -- busy fractions are of a launch alone on half the chip)
# Run on the GPU box from the repo root:  bash tools/pmc_passes.sh [key ...]   -> gpurun_out/pmc/<key>/<pass>/ + gpurun_out/pmc/<PMC_NAME: r04_pmc.json>
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc
mkdir -p $OUT
KEYS=${@:-"fused_m2_A256 gather_C4096_per_table_uniform gather_C4096_per_table_zipf gather_C4096_per_bank_uniform fused_h_B1024_bf16 gemm_C4096_f32 gemm_C4096_bf16 gemm_C4096_fp8 chain_gather_C4096_bank_bf16_img0 chain_gather_C4096_bank_bf16_img1 chain_gather_C4096_bank_fp8_img0 chain_gather_C4096_bank_fp8_img1"}
cd /tmp && export TMPDIR=/tmp
for key in $KEYS; do
  case $key in
    fused_m2_A256) ARGS="--legs roofline"; KERNEL="fr_fused_tile_m2_kernel"; EXTRA=("SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32");;
    gather_C4096_per_table_uniform) ARGS="--legs gather --gather-law uniform"; KERNEL="gather_pack"; EXTRA=();;
    gather_C4096_per_table_zipf) ARGS="--legs gather --gather-law zipf"; KERNEL="gather_pack"; EXTRA=();;
    gather_C4096_per_bank_uniform) ARGS="--legs bank"; KERNEL="gather_pack"; EXTRA=();;
    fused_h_B1024_bf16) ARGS="--roofline-only --model B --batch 1024 --precision bf16"; KERNEL="fr_fused_tile_h"; EXTRA=("SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE");;
    gemm_C4096_f32) ARGS="--roofline-only --model C --batch 4096 --precision f32"; KERNEL="fc_lp_gemm_kernel<0, 2"; EXTRA=("SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE");;
    gemm_C4096_bf16) ARGS="--roofline-only --model C --batch 4096 --precision bf16"; KERNEL="fc_pp_gemm_kernel<1, 3, 8"; EXTRA=("SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE");;
    gemm_C4096_fp8) ARGS="--roofline-only --model C --batch 4096 --precision fp8"; KERNEL="fc_pp_gemm_kernel<2, 2, 8"; EXTRA=("SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE");;
    chain_gather_C4096_bank_*)   # round 6: the in-chain gather of the per-bank chain, operand-type bank image on (img1) / off (img0); key = chain_gather_C4096_bank_<prec>_img<0|1>
      P=${key#chain_gather_C4096_bank_}; PRG="$ROOT/tools/experiments/lp_image_trace.py ${P%_img*} ${P#*_img} lone"; KERNEL="fr_pipeline_kernel<0,"; EXTRA=();;
    *) echo "unknown key $key"; exit 1;;
  esac
  [ -z "${PRG:-}" ] && PRG="$ROOT/bench.py --quick --no-gather-ab $ARGS"
  mkdir -p $OUT/$key
  for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "${EXTRA[@]}"; do
    tag=$(echo $pass | tr ' ' '_')
    timeout -k 10 240 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$key/$tag -- python3 $PRG > $OUT/$key/$tag.log 2>&1 || echo "pass $key/$tag failed"
  done
  python3 $ROOT/tools/pmc_summarize.py $OUT/$key $key "$KERNEL" $OUT/${PMC_NAME:-r04_pmc.json}
  PRG=""
done
cat $OUT/${PMC_NAME:-r04_pmc.json}
