#!/bin/bash
# per-kernel times of tools/attn_bench.py under rocprofv3 (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_attn
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_attn -o attn -- python tools/attn_variants.py > /tmp/attn_prof.log 2>&1
grep us /tmp/attn_prof.log
python - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/prof_attn/attn_kernel_stats.csv")))
for r in rows[:10]:
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")
    print("%9.1f us avg %5d calls  %s" % (float(r["AverageNs"])/1e3, int(r["Calls"]), n[:100]))
PY
