#!/bin/bash
# usage (GPU box): tools/pmc_gemm.sh > gpurun_out/<tag>_pmc_gemm.txt
# Matrix-core occupancy of the GEMM kernels on the path's shapes: SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) against
# SQ_BUSY_CYCLES (summed over the 32 shader engines' SQs) -> busy fraction = MFMA_BUSY / (32 * SQ_BUSY); LDS bank conflicts.
for only in fwd:2304x768 fwd:768x3072 dgrad:2304x768 dgrad:3072x768 wgrad:3072x768 wgrad:768x768; do
  echo "== $only"
  export GEMM_ONLY=$only GEMM_ITERS=5
  bash tools/pmc.sh "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" python tools/gemm_bench.py 2>&1 | grep -A4 "gemm_bf16_pp_kernel" | head -5
  python - <<PY
PY
done
