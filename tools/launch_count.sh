#!/bin/bash
# usage (GPU box, via gpurun): tools/launch_count.sh [tag]   -> gpurun_out/<tag>_launches_per_step.txt
# Kernel launches and kernel time PER STEP of the bench command, from the DIFFERENCE of two rocprofv3 --kernel-trace --stats runs
# (7 and 3 eager steps, side stream off): one-time work - parameter upload, momentum zero fills, the first step's 16-bit weight
# casts and transposes: ~530 launches - cancels.  (Round 3's "710 launches per step" divided a 5-step profile by 5 and so
# counted a fifth of that initialisation in every step.)
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in 3 7; do
  rm -rf /tmp/lc_$n
  EDITOR_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lc_$n -o lc -- \
    python bench.py --steps $n --warmup 0 --no-cpu-baseline --no-graph --no-replay --no-h2d --no-modes --no-eval > /tmp/lc_$n.log 2>&1
done
mkdir -p gpurun_out
python - <<PY > gpurun_out/${tag}_launches_per_step.txt
import csv
def load(n):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open("/tmp/lc_%d/lc_kernel_stats.csv" % n))}
a, b = load(3), load(7)
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    rows.append((k, (cb - ca) / 4.0, (tb - ta) / 4.0 / 1e6))
rows.sort(key=lambda r: -r[2])
gemm = lambda k: "gemm_bf16" in k or "slab_reduce" in k
ours = lambda k: "(anonymous namespace)" in k and "at::native" not in k or k.startswith("zero_tail")
print("per step (difference of a 7-step and a 3-step eager profile, side stream off):")
print("  launches %.1f   kernel ms %.3f   16-bit GEMM family: %.1f launches, %.3f ms   everything else: %.1f launches, %.3f ms" % (
    sum(r[1] for r in rows), sum(r[2] for r in rows), sum(r[1] for r in rows if gemm(r[0])), sum(r[2] for r in rows if gemm(r[0])),
    sum(r[1] for r in rows if not gemm(r[0])), sum(r[2] for r in rows if not gemm(r[0]))))
print("  torch / runtime kernels (not this repo's): %.1f launches, %.3f ms" % (
    sum(r[1] for r in rows if not ours(r[0])), sum(r[2] for r in rows if not ours(r[0]))))
print("  one-time launches that cancel: %d" % (sum(c for c, _ in a.values()) - 3 * sum(r[1] for r in rows)))
for k, n, ms in rows[:60]:
    if n or ms:
        print("%8.1f x %9.3f ms  %s" % (n, ms, k.replace("(anonymous namespace)::", "").replace("void ", "")[:130]))
PY
head -8 gpurun_out/${tag}_launches_per_step.txt
