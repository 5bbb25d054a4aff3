#!/bin/bash
# usage (GPU box): tools/pmc_mfma.sh > gpurun_out/<tag>_pmc_mfma.txt
# Matrix-core utilisation of the GEMM kernels INSIDE the training step (north_star: "rocprof showing ... MFMA utilisation for the
# GEMMs against MI355X peak"): one rocprofv3 --kernel-trace --pmc pass (counters only, no other trace domain) of the bench command,
#   util = SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (32 x SQ_BUSY_CYCLES (summed over the 32 shader engines))
# i.e. the share of the launch during which a SIMD's matrix pipe holds an MFMA, averaged over the chip; LDS bank conflicts beside it.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pmc_mfma
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv \
  -d /tmp/pmc_mfma -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-replay --no-h2d --no-modes --no-eval \
  > /tmp/pmc_mfma.log 2>&1
tail -1 /tmp/pmc_mfma.log | cut -c1-300
python - <<PY
import csv, collections, glob
f = glob.glob("/tmp/pmc_mfma/*counter_collection.csv")
if not f:
    print("no counter file", glob.glob("/tmp/pmc_mfma/*")); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:96]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
rows = []
for k, d in agg.items():
    if d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0 or d.get("SQ_BUSY_CYCLES", 0) <= 0:
        continue
    rows.append((d["SQ_VALU_MFMA_BUSY_CYCLES"], k, d, cnt[(k, "SQ_BUSY_CYCLES")]))
tm = sum(r[0] for r in rows); tb = sum(r[2]["SQ_BUSY_CYCLES"] for r in rows)
print("kernels with matrix-core work, 3 steps (2 timed + 1 warm-up) of the eager bench step, bf16, B = 128:")
print("%-98s %6s %10s %12s" % ("kernel", "n", "MFMA util", "LDS conflict"))
for _, k, d, n in sorted(rows, reverse=True):
    util = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (32.0 * d["SQ_BUSY_CYCLES"])
    conf = d.get("SQ_LDS_BANK_CONFLICT", 0) / max(d.get("SQ_LDS_IDX_ACTIVE", 0), 1)
    print("%-98s %6d %10.3f %12.3f" % (k, n, util, conf))
print("%-98s %6s %10.3f" % ("all of them (busy-cycle weighted)", "", tm / (32.0 * tb)))
PY
