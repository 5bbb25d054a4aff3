#!/bin/bash
# usage (on the GPU box, via gpurun): tools/prof.sh <tag> [bench args]   -> gpurun_out/prof_<tag>/kernel_stats.csv
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph "$@" > /tmp/bench_$tag.log 2>&1
tail -1 /tmp/bench_$tag.log | cut -c1-1400
mkdir -p gpurun_out/prof_$tag
cp /tmp/prof_$tag/${tag}_kernel_stats.csv gpurun_out/prof_$tag/kernel_stats.csv
python -c "import bench; print(bench.source_hash())" > gpurun_out/prof_$tag/kernel_stats.hash
python - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/prof_$tag/${tag}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms/step: %.2f" % (tot/5/1e6))
for r in rows[:32]:
    n=r["Name"]; n=n.replace("(anonymous namespace)::","").replace("void ","")
    print("%8.3f ms/step %5d calls/step %9.1f us avg  %s" % (float(r["TotalDurationNs"])/5/1e6, int(r["Calls"])//5, float(r["AverageNs"])/1e3, n[:110]))
PY
