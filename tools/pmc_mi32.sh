#!/bin/bash
# usage (GPU box): tools/pmc_mi32.sh > gpurun_out/<tag>_pmc_mi32.txt
# v_mfma_f32_32x32x16 against v_mfma_f32_16x16x32 in the SAME tile pipeline (EDITOR_PP_MI32, libeditor_gemm_mi32.so =
# `python -m editor_amd.build --mi32`), the fc1-shaped forward product with a bias-only and with the GELU epilogue: launch time beside
# matrix-core busy cycles, SQ busy cycles and LDS bank conflicts (one rocprofv3 --pmc pass per variant, counters only).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in default mi32; do
  if [ $lib = mi32 ]; then export GEMM_ALT_LIB=editor_amd/libeditor_gemm_mi32.so; else unset GEMM_ALT_LIB; fi
  for kind in none gelu; do
    python tools/gelu_lut_probe.py $kind | sed "s/^/  $lib /"
    rm -rf /tmp/pmc_mi32_$lib$kind
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv \
      -d /tmp/pmc_mi32_$lib$kind -o pmc -- python tools/gelu_lut_probe.py $kind > /tmp/pmc_mi32_$lib$kind.log 2>&1
    python - <<PY
import csv, collections, glob
f = glob.glob("/tmp/pmc_mi32_$lib$kind/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "gemm_bf16_pp_kernel" in k:
        agg[k.replace("(anonymous namespace)::", "").replace("void ", "")[:80]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    print("  %-8s %-6s %-76s MFMA busy %.3e  SQ busy %.3e  MFMA util %.3f  LDS conflict / active %.3f (%.3e / %.3e)" % ("$lib", "$kind", k,
          d["SQ_VALU_MFMA_BUSY_CYCLES"], d["SQ_BUSY_CYCLES"], d["SQ_VALU_MFMA_BUSY_CYCLES"] / (32.0 * max(d["SQ_BUSY_CYCLES"], 1)),
          d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1), d["SQ_LDS_BANK_CONFLICT"], d["SQ_LDS_IDX_ACTIVE"]))
PY
  done
done
