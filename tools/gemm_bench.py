"""Micro-benchmark of the bf16 GEMM kernel family on the hot path's real shapes (M = 3*128*129 token rows).
    python tools/gemm_bench.py            # prints TFLOP/s per (layout, shape)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editor_amd import ops  # noqa: E402
from editor_amd.functional import _splitk_for  # noqa: E402


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def use_trace_build():
    """EDITOR_GEMM_TRACE / EDITOR_GEMM_PP / EDITOR_GEMM_PP_STAGED are honoured by the DEBUG build of the GEMM only
    (libeditor_gemm_trace.so, include/editor_debug.h): build it and route the two GEMM entry points there."""
    from editor_amd import _lib, build
    build.build(trace=True)
    import ctypes
    tr = ctypes.CDLL(build.LIB_TRACE)
    lib = _lib.lib()
    for name in ("editor_gemm_bf16", "editor_gemm_f16"):
        fn = getattr(tr, name)
        fn.argtypes = lib.protos[name]
        fn.restype = ctypes.c_int
        lib._fn[name] = fn


def use_alt_build():
    """GEMM_ALT=1: route the two 16-bit GEMM entry points to libeditor_gemm_alt.so (the other K-tile choreography, editor_amd/build.py)."""
    from editor_amd import _lib, build
    import ctypes
    tr = ctypes.CDLL(build.LIB_ALT)
    lib = _lib.lib()
    for name in ("editor_gemm_bf16", "editor_gemm_f16"):
        fn = getattr(tr, name)
        fn.argtypes = lib.protos[name]
        fn.restype = ctypes.c_int
        lib._fn[name] = fn


def main():
    if os.environ.get("GEMM_ALT") == "1":
        use_alt_build()
    if any(os.environ.get(k) for k in ("EDITOR_GEMM_TRACE", "EDITOR_GEMM_PP", "EDITOR_GEMM_PP_STAGED")):
        use_trace_build()
    m = int(os.environ.get("GEMM_M", 3 * 128 * 129))
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rows = []
    shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]
    only = os.environ.get("GEMM_ONLY")          # e.g. "fwd:3072x768"
    if only:
        kind_only, sh = only.split(":")
        shapes = [tuple(int(v) for v in sh.split("x"))]
    iters = int(os.environ.get("GEMM_ITERS", "20"))
    for (n, k) in shapes:
        x = torch.randn(m, k, device=dev, generator=g).bfloat16()
        w = (torch.randn(n, k, device=dev, generator=g) * 0.05).bfloat16()
        dy = torch.randn(m, n, device=dev, generator=g).bfloat16()
        y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        dx = torch.empty(m, k, device=dev, dtype=torch.bfloat16)
        dw = torch.empty(n, k, device=dev)
        fl = 2.0 * m * n * k
        if not only or kind_only == "fwd":
            t = bench(lambda: ops.gemm(x, w, y, m, n, k, k, k, n, 0, 0), iters)
            rows.append(("fwd", n, k, fl / t / 1e9))
        if os.environ.get("GEMM_EPI") and (not only or kind_only == "fwd"):
            # the hot path's real forward epilogues: bias (qkv), bias + GELU + saved pre-activation (fc1),
            # bias + drop-path row scale + fp32 residual (proj / fc2)
            bias = torch.randn(n, device=dev, generator=g)
            t = bench(lambda: ops.gemm(x, w, y, m, n, k, k, k, n, 0, 0, bias=bias), iters)
            rows.append(("fwd+bias", n, k, fl / t / 1e9))
            pre = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
            # as the step runs it: the forward saves gelu'(pre-activation) for the backward
            t = bench(lambda: ops.gemm(x, w, y, m, n, k, k, k, n, 0, 0, bias=bias, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD,
                                       aux=pre), iters)
            rows.append(("fwd+gelu", n, k, fl / t / 1e9))
            res = torch.randn(m, n, device=dev, generator=g)
            yf = torch.empty(m, n, device=dev)
            rs = torch.rand(m, device=dev, generator=g)
            t = bench(lambda: ops.gemm(x, w, yf, m, n, k, k, k, n, 0, 0, bias=bias, rowscale=rs, epilogue=ops.EPI_RESIDUAL, aux=res), iters)
            rows.append(("fwd+resid", n, k, fl / t / 1e9))
            dpre = torch.empty(m, k, device=dev, dtype=torch.bfloat16)
            prek = torch.randn(m, k, device=dev, generator=g).bfloat16()
            # as the training step runs it: aux = gelu'(pre-activation) saved by the forward, B = the k-major W^T copy
            wt = w.t().contiguous()
            t = bench(lambda: ops.gemm(dy, wt, dpre, m, k, n, n, n, k, 0, 0, epilogue=ops.EPI_GELU_BWD | ops.EPI_AUX_GRAD,
                                       aux=prek), iters)
            rows.append(("dgrad+gelu'", n, k, fl / t / 1e9))
        if not only or kind_only == "dgrad":
            wt = w.t().contiguous()
            t = bench(lambda: ops.gemm(dy, wt, dx, m, k, n, n, n, k, 0, 0), iters)
            rows.append(("dgrad", n, k, fl / t / 1e9))
            t = bench(lambda: ops.gemm(dy, w, dx, m, k, n, n, k, k, 0, 1), iters)
            rows.append(("dgrad(row-k W)", n, k, fl / t / 1e9))
        sk, skf = _splitk_for(n, k, m)
        if not only or kind_only == "wgrad":
            t = bench(lambda: ops.gemm(dy, x, dw, n, k, m, n, k, k, 1, 1, splitk=sk, epilogue=skf), iters)
            rows.append((f"wgrad(sk={sk})", n, k, fl / t / 1e9))
    if os.environ.get("GEMM_SPLIT"):
        # the split-precision forward products (COMPUTE_DTYPE 'f16x2'): three half products per output, rate quoted on 3x the FLOPs
        for (n, k) in shapes:
            x = ops.split_f32(torch.randn(m, k, device=dev, generator=g))
            w = ops.split_f32(torch.randn(n, k, device=dev, generator=g) * 0.05, ops.SPLIT_WSCALE)
            bias = torch.randn(n, device=dev, generator=g)
            fl = 3 * 2.0 * m * n * k
            yh, yl = torch.empty(m, n, device=dev, dtype=torch.float16), torch.empty(m, n, device=dev, dtype=torch.float16)
            t = bench(lambda: ops.gemm_split(x, w, yh, yl, m, n, k, alpha=1.0 / ops.SPLIT_WSCALE, bias=bias), iters)
            rows.append(("x2 fwd pair", n, k, fl / t / 1e9))
            yf = torch.empty(m, n, device=dev)
            t = bench(lambda: ops.gemm_split(x, w, yf, None, m, n, k, alpha=1.0 / ops.SPLIT_WSCALE, bias=bias), iters)
            rows.append(("x2 fwd fp32", n, k, fl / t / 1e9))
            aux = torch.empty(m, n, device=dev, dtype=torch.float16)
            t = bench(lambda: ops.gemm_split(x, w, yh, yl, m, n, k, alpha=1.0 / ops.SPLIT_WSCALE, bias=bias,
                                             epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD, aux=aux), iters)
            rows.append(("x2 fwd+gelu", n, k, fl / t / 1e9))
            res = torch.randn(m, n, device=dev, generator=g)
            rs = torch.rand(m, device=dev, generator=g)
            t = bench(lambda: ops.gemm_split(x, w, yf, None, m, n, k, alpha=1.0 / ops.SPLIT_WSCALE, bias=bias, rowscale=rs,
                                             epilogue=ops.EPI_RESIDUAL, aux=res), iters)
            rows.append(("x2 fwd+resid", n, k, fl / t / 1e9))
    for r in rows:
        print("%-14s N=%-5d K=%-5d %8.1f TFLOP/s" % r)


if __name__ == "__main__":
    main()
