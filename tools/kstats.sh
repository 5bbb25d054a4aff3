#!/bin/bash
# usage (GPU box): tools/kstats.sh <cmd...>   -> per-kernel average durations of the command (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/kstats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -o k -- "$@" > /tmp/kstats.log 2>&1
python - <<PY
import csv
rows = list(csv.DictReader(open("/tmp/kstats/k_kernel_stats.csv")))
for r in rows[:16]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%6d calls %10.1f us avg  %5.1f %%  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]), n[:100]))
PY
