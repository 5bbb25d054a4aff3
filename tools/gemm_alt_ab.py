"""A/B of the ping-pong kernel's K-tile choreography: libeditor_hip.so (EDITOR_PP_PHASES default) against libeditor_gemm_alt.so (the
other one, `python -m editor_amd.build --alt`) on the hot path's products with their real epilogues, M = 3*128*129 token rows:
bit-for-bit equality of every output, then us per launch of both (operands rotating over NSETS sets).
    python tools/gemm_alt_ab.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editor_amd import _lib, build, ops  # noqa: E402
from editor_amd.functional import _splitk_for  # noqa: E402

NAMES = ("editor_gemm_bf16", "editor_gemm_f16", "editor_gemm_wgrad_group", "editor_gemm_group", "editor_gemm_f16x2")


def route(alt):
    lib = _lib.lib()
    src = ctypes.CDLL(os.environ.get("GEMM_ALT_LIB", build.LIB_ALT)) if alt else lib.cdll
    for name in NAMES:
        fn = getattr(src, name)
        fn.argtypes = lib.protos[name]
        fn.restype = ctypes.c_int
        lib._fn[name] = fn


def bench(fns, iters=24):
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
    m = int(os.environ.get("GEMM_M", 3 * 128 * 129))
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    nsets = int(os.environ.get("NSETS", "3"))
    dt = torch.float16 if os.environ.get("GEMM_DTYPE") == "f16" else torch.bfloat16
    th768 = ops.EPI_TILE_ROWS(ops.gemm_tile_rows(m, 768))
    prods = [("qkv fwd+bias", 2304, 768, "bias"), ("fc1 fwd+gelu", 3072, 768, "gelu"), ("proj fwd+resid", 768, 768, "resid"),
             ("fc2 fwd+resid", 768, 3072, "resid"), ("fc2 dgrad+gelu'", 3072, 768, "gelu_bwd"), ("fc1 dgrad", 768, 3072, "plain"),
             ("qkv dgrad", 768, 2304, "plain"), ("proj dgrad", 768, 768, "plain"), ("fc1 wgrad", 3072, 768, "wgrad"),
             ("proj wgrad", 768, 768, "wgrad"), ("block wgrad group", 0, 0, "wgroup")]
    print("%-20s %10s %10s %8s   %s" % ("product", "default us", "alt us", "alt/def", "bits"))
    tot = [0.0, 0.0]
    for name, n, k, kind in prods:
        sets = []
        for _ in range(nsets):
            if kind == "wgroup":
                x768 = torch.randn(m, 768, device=dev, generator=g).to(dt)
                x3072 = torch.randn(m, 3072, device=dev, generator=g).to(dt)
                dy2304 = torch.randn(m, 2304, device=dev, generator=g).to(dt)
                jobs = [(dy2304, x768, torch.empty(2304, 768, device=dev)), (x768, x768, torch.empty(768, 768, device=dev)),
                        (x3072, x768, torch.empty(3072, 768, device=dev)), (x768, x3072, torch.empty(768, 3072, device=dev))]
                sets.append(jobs)
                continue
            x = torch.randn(m, k, device=dev, generator=g).to(dt)
            w = (torch.randn(n, k, device=dev, generator=g) * 0.05).to(dt)
            bias = torch.randn(n, device=dev, generator=g)
            y = torch.empty(m, n, device=dev, dtype=dt)
            aux = torch.randn(m, n, device=dev, generator=g).to(dt)
            res = torch.randn(m, n, device=dev, generator=g) if kind == "resid" else None
            yf = torch.empty(m, n, device=dev) if kind == "resid" else None
            rs = torch.rand(m, device=dev, generator=g)
            dy = torch.randn(m, n, device=dev, generator=g).to(dt) if kind == "wgrad" else None
            dw = torch.empty(n, k, device=dev) if kind == "wgrad" else None
            sets.append((x, w, bias, y, aux, res, yf, rs, dy, dw))

        def mk(s_):
            if kind == "wgroup":
                return (lambda: ops.gemm_wgrad_group(s_, m)), [j[2] for j in s_]
            x, w, bias, y, aux, res, yf, rs, dy, dw = s_
            if kind == "bias":
                return (lambda: ops.gemm(x, w, y, m, n, k, k, k, n, 0, 0, bias=bias)), [y]
            if kind == "gelu":
                return (lambda: ops.gemm(x, w, y, m, n, k, k, k, n, 0, 0, bias=bias, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD, aux=aux)), [y, aux]
            if kind == "resid":
                return (lambda: ops.gemm(x, w, yf, m, n, k, k, k, n, 0, 0, bias=bias, rowscale=rs, epilogue=ops.EPI_RESIDUAL | th768, aux=res)), [yf]
            if kind == "gelu_bwd":
                return (lambda: ops.gemm(x, w, y, m, n, k, k, k, n, 0, 0, epilogue=ops.EPI_GELU_BWD | ops.EPI_AUX_GRAD, aux=aux)), [y]
            if kind == "wgrad":
                sk, skf = _splitk_for(n, k, m)
                return (lambda: ops.gemm(dy, x, dw, n, k, m, n, k, k, 1, 1, splitk=sk, epilogue=skf)), [dw]
            return (lambda: ops.gemm(x, w, y, m, n, k, k, k, n, 0, 0, epilogue=(th768 if n == 768 else 0))), [y]
        made = [mk(s_) for s_ in sets]
        # bits: run set 0 with both libraries
        route(False); made[0][0](); torch.cuda.synchronize(); ref = [o.clone() for o in made[0][1]]
        for o in made[0][1]:
            o.zero_()
        route(True); made[0][0](); torch.cuda.synchronize()
        same = all(torch.equal(a_, b_) for a_, b_ in zip(ref, made[0][1]))
        note = "identical"
        if not same:       # (EDITOR_PP_MI32: 16 k per MFMA instead of 32 - another fp32 summation order; how far apart, how many elements)
            rel = max(((a_.double() - b_.double()).norm() / a_.double().norm().clamp_min(1e-30)).item() for a_, b_ in zip(ref, made[0][1]))
            frac = max((a_ != b_).double().mean().item() for a_, b_ in zip(ref, made[0][1]))
            note = "DIFFER rel-L2 %.2e, %.3f %% of the elements" % (rel, 100 * frac)
        route(False); t0 = bench([f for f, _ in made])
        route(True); t1 = bench([f for f, _ in made])
        route(False); t0b = bench([f for f, _ in made])
        route(True); t1b = bench([f for f, _ in made])
        t0, t1 = min(t0, t0b), min(t1, t1b)
        tot[0] += t0; tot[1] += t1
        print("%-20s %10.1f %10.1f %8.3f   %s" % (name, t0, t1, t1 / t0, note))
        del sets, made
        torch.cuda.empty_cache()
    print("%-20s %10.1f %10.1f %8.3f" % ("sum", tot[0], tot[1], tot[1] / tot[0]))


if __name__ == "__main__":
    main()
