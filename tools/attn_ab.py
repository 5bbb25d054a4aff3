"""Same-box A/B of round 6's two padding-tile switches of csrc/attention_bf16.hip: libeditor_hip.so (ATTN_ROLLOUT_SKIP = 1,
ATTN_PAIR_SKIP = 0: what ships) against libeditor_attn_alt.so (`python -m editor_amd.build --attn-alt`: the same source with BOTH
flipped) on the backbone's shape - 384 sequences x 12 heads x 129 tokens x 64 - bit-for-bit equality of every output, then us per
launch of both, alternating, operands rotating over NSETS sets (> the Infinity Cache).  So the forward / backward rows read
"no pair skip (product) vs pair skip", the rollout row "skip (product) vs no skip".
    python tools/attn_ab.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editor_amd import _lib, build, ops  # noqa: E402

NAMES = ("editor_attention_fwd_bf16", "editor_attention_bwd_bf16", "editor_attention_bwd_colsum_bf16", "editor_attn_rollout_step_bf16")


def route(alt):
    lib = _lib.lib()
    src = ctypes.CDLL(build.LIB_ATTN_ALT) if alt else lib.cdll
    for name in NAMES:
        fn = getattr(src, name)
        fn.argtypes = lib.protos[name]
        fn.restype = ctypes.c_int
        lib._fn[name] = fn


def bench(fns, iters=30):
    for f in fns:
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(iters):
        fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    t = int(os.environ.get("ATTN_T", "129"))
    b, heads, hd = 384, 12, 64
    nsets = int(os.environ.get("NSETS", "3"))
    g = torch.Generator(device="cuda").manual_seed(0)
    sets = []
    for _ in range(nsets):
        qkv = (torch.randn(b * t, 3 * heads * hd, device="cuda", generator=g) * 0.5).bfloat16()
        do = torch.randn(b * t, heads * hd, device="cuda", generator=g).bfloat16()
        sets.append((qkv, do))
    route(False)
    saved = [ops.attention_fwd(q, b, t, heads, hd, None, None) for q, _ in sets]
    r_in = torch.rand(b * heads, t, device="cuda", generator=g)

    def step(i, kind):
        qkv, do = sets[i]
        o, lse = saved[i]
        if kind == "fwd":
            return ops.attention_fwd(qkv, b, t, heads, hd, None, None)
        if kind == "bwd":
            cs = torch.empty(3 * heads * hd, device="cuda")
            return ops.attention_bwd(qkv, do, b, t, heads, hd, None, lse, o, colsum=cs), cs
        return ops.attn_rollout_qk([(qkv, lse), (qkv, lse)], b, t, heads, hd)       # first (one-hot) step + one dense step

    def flat(x):
        out = []
        for v in (x if isinstance(x, (tuple, list)) else (x,)):
            out += flat(v) if isinstance(v, (tuple, list)) else [v]
        return [v for v in out if isinstance(v, torch.Tensor)]

    print("T = %d   %-22s %10s %10s %8s   %s" % (t, "kernel(s)", "product us", "alt us", "ratio", "bits"))
    for kind, label in (("fwd", "forward"), ("bwd", "backward (dq, dk/dv)"), ("roll", "rollout, 2 layers")):
        route(False); a = flat(step(0, kind)); torch.cuda.synchronize(); a = [v.clone() for v in a]
        route(True); b_ = flat(step(0, kind)); torch.cuda.synchronize()
        same = len(a) == len(b_) and all(torch.equal(x, y) for x, y in zip(a, b_))
        fns = [lambda i=i: step(i, kind) for i in range(nsets)]
        ts = [[], []]
        for rep in range(4):                               # A B B A A B B A: whichever build runs second in a pair reads ~2 % faster
            for alt in ((False, True) if rep % 2 == 0 else (True, False)):
                route(alt)
                ts[alt].append(bench(fns))
        route(False)
        t0, t1 = min(ts[0]), min(ts[1])
        print("          %-22s %10.1f %10.1f %8.3f   %s   (runs: %s | %s)" % (label, t0, t1, t0 / t1, "identical" if same else "DIFFER",
              " ".join("%.1f" % v for v in ts[0]), " ".join("%.1f" % v for v in ts[1])))


if __name__ == "__main__":
    main()
