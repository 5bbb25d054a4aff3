#!/bin/bash
# usage (GPU box): tools/pmc_lds_conflicts.sh > gpurun_out/<tag>_pmc_gelu_lut.txt
# LDS bank conflicts of the fc1 forward product with and without its GELU table: one rocprofv3 --pmc pass per
# variant (counters only), SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE and the matrix-core utilisation beside the launch time.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for kind in none gelu gelu_nosave gelu_f16; do
  python tools/gelu_lut_probe.py $kind
  rm -rf /tmp/pmc_lut_$kind
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv \
    -d /tmp/pmc_lut_$kind -o pmc -- python tools/gelu_lut_probe.py $kind > /tmp/pmc_lut_$kind.log 2>&1
  python - <<PY
import csv, collections, glob
f = glob.glob("/tmp/pmc_lut_$kind/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "gemm_bf16_pp_kernel" in k:
        agg[k.replace("(anonymous namespace)::", "").replace("void ", "")[:80]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    print("  %-10s %-80s LDS conflict / active %.3f   conflict cycles %.3e   active %.3e   MFMA util %.3f" % ("$kind", k,
          d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1), d["SQ_LDS_BANK_CONFLICT"], d["SQ_LDS_IDX_ACTIVE"],
          d["SQ_VALU_MFMA_BUSY_CYCLES"] / (32.0 * max(d["SQ_BUSY_CYCLES"], 1))))
PY
done
