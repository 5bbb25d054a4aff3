#!/bin/bash
# usage (on the GPU box, via gpurun): [TAG=r03] tools/pmc_traffic.sh   -> gpurun_out/<TAG>_pmc_traffic.json (+ .txt)
# HBM-side traffic of the 16-bit GEMM family in the bench command, as MI355X_MICROARCH.md "HBM" prescribes: separate
# --pmc passes for FETCH_SIZE and WRITE_SIZE (they do not fit one pass), --kernel-trace only (no other trace domains),
# unit KB, FETCH_SIZE doubled on gfx950 (128-byte requests tallied at 64 B).  Infinity-Cache hits are counted by these
# memory-side request counters, so the figure is an upper bound on HBM bytes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${TAG:-r04}
STEPS=2; WARM=1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$ctr -o pmc -- \
    python bench.py --steps $STEPS --warmup $WARM --no-graph --no-cpu-baseline --no-replay --no-h2d --no-modes --no-eval > /tmp/pmc_$ctr.log 2>&1
  tail -1 /tmp/pmc_$ctr.log | cut -c1-300
done
mkdir -p gpurun_out
python - <<PY
import csv, glob, json, collections
steps = $STEPS + $WARM
out = {}
txt = []
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/pmc_%s/*counter_collection.csv" % ctr)
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != ctr:
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    out[ctr] = (agg, cnt)
    txt.append("== %s (KB; per-dispatch average; %d steps incl. warm-up)" % (ctr, steps))
    for k in sorted(agg, key=lambda k: -agg[k])[:24]:
        txt.append("%14.0f KB total %10.1f KB/dispatch  n=%-5d %s" % (agg[k], agg[k] / cnt[k], cnt[k], k[:120]))
gemm = lambda k: "gemm_bf16_p" in k or "gemm_bf16_kernel" in k or "slab_reduce" in k
fetch = 2.0 * 1024 * sum(v for k, v in out["FETCH_SIZE"][0].items() if gemm(k)) / steps      # gfx950: x2, KB -> bytes
write = 1024.0 * sum(v for k, v in out["WRITE_SIZE"][0].items() if gemm(k)) / steps
launches = sum(v for k, v in out["FETCH_SIZE"][1].items() if gemm(k)) / steps
# algorithmic bytes of the same launches: operands read once + output written once (+ fp32 residual / saved pre-activation)
M, D, H = 3 * 128 * 129, 768, 3072
blocks = 12
fwd = blocks * ((M * D + 3 * D * D) * 2 + M * 3 * D * 2          # qkv
                + (M * D + D * D) * 2 + M * D * 8                # proj (+ fp32 residual read, fp32 out)
                + (M * D + D * H) * 2 + 2 * M * H * 2            # fc1 (+ pre-activation and activation out)
                + (M * H + D * H) * 2 + M * D * 8)               # fc2
dgrad = blocks * ((M * D + D * H) * 2 + M * H * 2 + M * H * 2    # fc2 dgrad (+ saved pre-activation)
                  + (M * H + D * H) * 2 + M * D * 2              # fc1 dgrad
                  + (M * D + D * D) * 2 + M * D * 2              # proj dgrad
                  + (M * 3 * D + 3 * D * D) * 2 + M * D * 2)     # qkv dgrad
wgrad = blocks * ((M * D + M * H) * 2 * 2 + 2 * D * H * 4 + (M * D * 2) * 2 + D * D * 4 + (M * 3 * D + M * D) * 2 + 3 * D * D * 4)
alg = fwd + dgrad + wgrad
res = {"gemm_hbm_bytes_per_step": int(fetch + write), "gemm_fetch_bytes_per_step": int(fetch), "gemm_write_bytes_per_step": int(write),
       "gemm_launches_per_step": launches, "gemm_alg_bytes_per_step": int(alg),
       "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of the eager bench command (bench.py --no-graph), GEMM-family "
               "kernels summed per step; FETCH_SIZE x2 (gfx950) and KB->bytes; memory-side request counters include Infinity-Cache "
               "hits (upper bound on HBM bytes); alg bytes = backbone blocks only (operands once + outputs once)"}
json.dump(res, open("gpurun_out/$TAG" + "_pmc_traffic.json", "w"), indent=1)
open("gpurun_out/$TAG" + "_pmc_traffic.txt", "w").write("\n".join(txt) + "\n" + json.dumps(res, indent=1) + "\n")
print(json.dumps(res))
PY
