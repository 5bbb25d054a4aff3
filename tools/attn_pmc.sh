#!/bin/bash
# SQ wave-time breakdown of the attention kernels (tools/attn_bench.py) - one --pmc pass, --kernel-trace only
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pmc_attn
rocprofv3 --kernel-trace --pmc ${ATTN_PMC:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES} --output-format csv -d /tmp/pmc_attn -o pmc -- python tools/attn_variants.py > /tmp/pmc_attn.log 2>&1
tail -2 /tmp/pmc_attn.log
python - <<PY
import csv, collections, glob
f = glob.glob("/tmp/pmc_attn/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::","")[:60]
    if "attn" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    wc = d.get("SQ_WAVE_CYCLES", 1)
    print(k)
    print("    " + "  ".join("%s=%.3f" % (c.replace("SQ_",""), v / wc) for c, v in d.items() if c != "SQ_WAVE_CYCLES"), " wave_cycles/dispatch=%.3e" % (wc / cnt[(k,"SQ_WAVE_CYCLES")]))
PY
