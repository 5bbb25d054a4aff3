"""Rewrites the GENERATED blocks of DESIGN.md / README.md from the committed measurement files, so that text and evidence cannot
disagree ("30.4 vs 30.84").

    python tools/doc_numbers.py            # rewrite the blocks in place
    python tools/doc_numbers.py --check    # exit 1 if a block differs from what the files give (tests/test_host.py runs this)

A block is everything between `<!-- BEGIN GENERATED: <name> -->` and `<!-- END GENERATED: <name> -->`.  Sources (profiles/):
    r06_bench_line_default.json        the driver's command (`python bench.py`), builder-run on one MI355X
    r06_bench_line_noskip.json         the same with EDITOR_DROP_SKIP=0 (stochastic-depth compaction off), same box, same call
    r06_bench_line_{f16,f16x2,f16x2s,rgbnt100,msvr310,synth4l}.json
    r06_bench_kernel_stats_serial.csv  rocprofv3 --kernel-trace --stats of the bench command, weight gradients on the main stream
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
TAG = "r06"


def _line(name):
    f = os.path.join(P, "%s_bench_line_%s.json" % (TAG, name))
    if not os.path.exists(f):
        return None
    txt = open(f).read().strip()
    return json.loads(txt.splitlines()[-1]) if txt else None


def _csv(name):
    f = os.path.join(P, "%s_%s.csv" % (TAG, name))
    if not os.path.exists(f):
        return None
    return list(csv.DictReader(open(f)))


def _clean(n):
    return n.replace("void ", "").replace("(anonymous namespace)::", "")


def serial_summary():
    rows = _csv("bench_kernel_stats_serial")
    if not rows:
        return None
    steps = None
    for r in rows:
        if "sgd_multi_kernel" in r["Name"]:
            steps = int(r["Calls"])
    steps = steps or 5
    tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
    fam = [r for r in rows if "gemm_bf16_pp" in r["Name"] or "gemm_bf16_pipe" in r["Name"] or "gemm_bf16_kernel" in r["Name"]]
    gemm = sum(float(r["TotalDurationNs"]) for r in fam) / steps / 1e6
    slab = sum(float(r["TotalDurationNs"]) for r in rows if "slab_reduce" in r["Name"]) / steps / 1e6
    launches = sum(int(r["Calls"]) for r in rows) / steps
    top = sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]
    return dict(steps=steps, total=tot, gemm=gemm, slab=slab, launches=launches,
                top=[(_clean(r["Name"])[:78], float(r["TotalDurationNs"]) / steps / 1e6, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3)
                     for r in top])


def numbers_block():
    d = _line("default")
    out = []
    if d is None:
        return "(no profiles/%s_bench_line_default.json yet)\n" % TAG
    r = d["roofline"]
    out.append("| figure | value | source |")
    out.append("|---|---|---|")
    out.append("| headline: config 2, bf16, B = 128, one MI355X, the reference loop's definition (H2D of every batch + a sync per iteration) | "
               "**%.0f img/s = %.2f ms per step** | `profiles/%s_bench_line_default.json` (`value`, `ms_per_step`) |" % (d["value"], d["ms_per_step"], TAG))
    if d.get("replay_only"):
        out.append("| the bare step (inputs resident, one sync after K steps) | %.0f img/s = %.2f ms | same file, `replay_only` |"
                   % (d["replay_only"]["value"], d["replay_only"]["ms_per_step"]))
    ns = _line("noskip")
    if ns:
        out.append("| the same command with the stochastic-depth compaction off (`EDITOR_DROP_SKIP=0`), same box, same gpurun call | "
                   "%.0f img/s = %.2f ms (GEMM family in situ %.2f ms) | `profiles/%s_bench_line_noskip.json` |"
                   % (ns["value"], ns["ms_per_step"], ns["roofline"]["gemm_ms_per_step"], TAG))
    out.append("| `roofline.frac`: the reference's ALGORITHMIC FLOPs of the 16-bit GEMM family (%.2f TF per step) / its time INSIDE the step "
               "(%.2f ms, HIP events around every launch of two eager steps, mean) | **%.1f TFLOP/s = %.4f of the 2.5 PF dense peak** | `roofline.achieved`, `.frac`, "
               "`.gemm_ms_per_step` |" % (r["alg_tflop_per_step"], r["gemm_ms_per_step"], r["achieved"], r["frac"]))
    if r.get("executed_tflop_per_step") is not None:
        out.append("| the FLOPs the launches really executed (the dropped samples' MLP rows are not computed) | %.2f TF per step -> %.1f TFLOP/s = %.4f | "
                   "`roofline.executed_tflop_per_step`, `.achieved_executed`, `.frac_executed` |"
                   % (r["executed_tflop_per_step"], r["achieved_executed"], r["frac_executed"]))
    bk = r.get("by_kind_in_situ") or {}
    if bk:
        out.append("| by kind, in situ (algorithmic TFLOP/s; ms per step; launches) | " +
                   "; ".join("%s %.0f (%.2f ms, %d)" % (k, v["tflops"], v["ms_per_step"], v["launches"]) for k, v in bk.items()) +
                   " | `roofline.by_kind_in_situ` |")
    out.append("| the same launches replayed back to back | %.1f TFLOP/s = %.4f | `roofline.achieved_replay`, `.frac_replay` |"
               % (r["achieved_replay"], r["frac_replay"]))
    out.append("| whole step: SURVEY 8(d)'s %.2f TF / `ms_per_step` | %.4f of peak | `roofline.step_frac` |" % (r["step_alg_tflop"], r["step_frac"]))
    if r.get("non_gemm_in_situ"):
        ng = r["non_gemm_in_situ"]
        out.append("| everything that is not a 16-bit GEMM, in situ (per library entry, same eager steps) | %.2f ms per step: %s | "
                   "`roofline.non_gemm_in_situ` |" % (ng["ms_per_step"], ", ".join("`%s` %.2f" % (k.replace("editor_", ""), v[0])
                                                                                  for k, v in list(ng["top"].items())[:7])))
    s = serial_summary()
    if s:
        out.append("| rocprofv3 `--kernel-trace --stats` of the bench command, weight gradients on the main stream (%d steps, one-time "
                   "launches included) | %.2f ms of kernels per step; GEMM kernels %.2f ms (+ %.2f ms slab folds) -> %.0f TFLOP/s = %.3f "
                   "(%.3f with the folds) | `profiles/%s_bench_kernel_stats_serial.csv` |"
                   % (s["steps"], s["total"], s["gemm"], s["slab"], r["alg_tflop_per_step"] / s["gemm"] * 1e3,
                      r["alg_tflop_per_step"] / s["gemm"] * 1e3 / 2500.0, r["alg_tflop_per_step"] / (s["gemm"] + s["slab"]) * 1e3 / 2500.0, TAG))
    lf = os.path.join(P, "%s_launches_per_step.txt" % TAG)
    if os.path.exists(lf):
        m = re.search(r"launches\s+([\d.]+)\s+kernel ms\s+([\d.]+)\s+16-bit GEMM family:\s+([\d.]+) launches,\s+([\d.]+) ms\s+everything else:\s+([\d.]+) launches,\s+([\d.]+) ms",
                      open(lf).read())
        if m:
            out.append("| launches and kernel time per step, by difference of a 7-step and a 3-step eager profile | %s launches, %s ms of kernels = "
                       "%s ms in %s GEMM-family launches + %s ms in %s others | `profiles/%s_launches_per_step.txt` |"
                       % (m.group(1), m.group(2), m.group(4), m.group(3), m.group(6), m.group(5), TAG))
    for k in r.get("hbm_kernels", []):
        pass
    if r.get("hbm_kernels"):
        out.append("| memory-bound kernels, rotating operand sets > 512 MB (fraction of 8 TB/s; in situ) | " +
                   "; ".join("%s %.3f (%s)" % (k["kernel"].split(" ")[0].replace("_kernel", ""), k["frac"],
                                               ("%.3f" % k["in_situ_frac"]) if "in_situ_frac" in k else "-") for k in r["hbm_kernels"]) +
                   " | `roofline.hbm_kernels` |")
    modes = d.get("modes") or {}
    rows = []
    for m in ("bf16", "f16", "f16x2s", "f16x2", "f32"):
        e = modes.get(m)
        if e and e.get("value"):
            rows.append("%s %.0f img/s (cls4t %.1e, index %s)" % (m, e["value"], e["cls4t_rel_err"], "identical" if e["index_bit_identical"] else "differs"))
    if rows:
        out.append("| every compute mode in the same timed loop, accuracy on a B = 16 eval sample vs the oracle | " + "; ".join(rows) + " | `modes` |")
    if d.get("value_at_parity"):
        vp = d["value_at_parity"]
        out.append("| fastest mode meeting BOTH north-star criteria (indices bit-identical, features <= 1e-3) | **%s: %.0f img/s** | `value_at_parity` |"
                   % (vp["mode"], vp["value"]))
    for name, label in (("f16", "f16"), ("f16x2s", "f16x2s"), ("f16x2", "f16x2"), ("rgbnt100", "config 3 (RGBNT100, one GPU)"),
                        ("msvr310", "config 4 (384x128, one GPU)"), ("synth4l", "config 5 (4-modal ViT-L, 32 per GPU)")):
        e = _line(name)
        if e:
            out.append("| `bench.py` %s | %.0f %s = %.2f ms per step | `profiles/%s_bench_line_%s.json` |"
                       % (label, e["value"], e["unit"].replace("tri-modal ", ""), e["ms_per_step"], TAG, name))
    oc = d.get("other_configs")
    if oc:
        parts = ["%s %s: %.0f %s (%.2f ms)" % (v["baseline_config"], k, v["value"], v["unit"].replace("tri-modal ", ""), v["ms_per_step"])
                 for k, v in oc.items() if isinstance(v, dict) and v.get("value")]
        if parts:
            out.append("| the other BASELINE configs inside the default line's own run (child process each, same box) | " + "; ".join(parts)
                       + " | `other_configs` |")
    cb = d.get("cpu_baseline")
    if cb:
        out.append("| CPU baseline (the oracle on the GPU box's host cores, same run) | %.2f %s on %s cores (%s) | `cpu_baseline` |"
                   % (cb["value"], cb["unit"], cb["cores"], cb.get("kind", "port")))
    return "\n".join(out) + "\n"


def serial_block():
    s = serial_summary()
    if not s:
        return "(no profiles/%s_bench_kernel_stats_serial.csv yet)\n" % TAG
    out = ["| kernel | ms per step | launches per step | average us |", "|---|---|---|---|"]
    for n, ms, calls, us in s["top"]:
        out.append("| `%s` | %.3f | %.0f | %.1f |" % (n.replace("|", "/"), ms, calls, us))
    out.append("| all kernels (one-time launches of the %d profiled steps included) | %.2f | %.0f | |" % (s["steps"], s["total"], s["launches"]))
    return "\n".join(out) + "\n"


BLOCKS = {"r06-numbers": numbers_block, "r06-serial": serial_block}


def rewrite(path, check):
    txt = open(path).read()
    new = txt
    for name, fn in BLOCKS.items():
        pat = re.compile(r"(<!-- BEGIN GENERATED: %s -->\n)(.*?)(<!-- END GENERATED: %s -->)" % (re.escape(name), re.escape(name)), re.S)
        if pat.search(new):
            body = fn()
            new = pat.sub(lambda m: m.group(1) + body + m.group(3), new)
    if new != txt:
        if check:
            return False
        open(path, "w").write(new)
    return True


def main():
    check = "--check" in sys.argv
    ok = True
    for f in ("DESIGN.md", "README.md"):
        ok &= rewrite(os.path.join(ROOT, f), check)
    if check and not ok:
        print("generated blocks of DESIGN.md / README.md are stale: run python tools/doc_numbers.py")
        sys.exit(1)


if __name__ == "__main__":
    main()
