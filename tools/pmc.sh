#!/bin/bash
# usage: tools/pmc.sh "<counters>" <cmd...>   (PMC pass; prints per-kernel sums)
ctrs="$1"; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pmc_out
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_out -o pmc -- "$@" > /tmp/pmc_cmd.log 2>&1
tail -3 /tmp/pmc_cmd.log
python - <<PY
import csv, collections, glob
f = glob.glob("/tmp/pmc_out/*counter_collection.csv")
if not f:
    print("no counter file", glob.glob("/tmp/pmc_out/*")); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::","")[:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k)
    for c, v in d.items():
        print("    %-28s %16.0f  (per dispatch %14.0f, n=%d)" % (c, v, v / cnt[(k, c)], cnt[(k, c)]))
PY
