"""Sweep of EDITOR_EPI_STAGGER on the hot path's forward / dgrad products with their real epilogues (M = 3*128*129 token rows).
    python tools/stagger_sweep.py            # us per launch per (product, stagger in units of 2048 cycles)
Operands rotate over `NSETS` sets so that the row operand / residual of a launch is not the previous launch's (Infinity Cache)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editor_amd import ops  # noqa: E402


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
    levels = [int(v) for v in os.environ.get("STAGGER", "0,8,16,24,32,48,63").split(",")]
    th768 = ops.EPI_TILE_ROWS(ops.gemm_tile_rows(m, 768))
    prods = [  # name, n, k, kind
        ("qkv fwd+bias", 2304, 768, "bias"), ("fc1 fwd+gelu", 3072, 768, "gelu"), ("proj fwd+resid", 768, 768, "resid"),
        ("fc2 fwd+resid", 768, 3072, "resid"), ("fc2 dgrad+gelu'", 3072, 768, "gelu_bwd"), ("fc1 dgrad", 768, 3072, "plain"),
        ("qkv dgrad", 768, 2304, "plain"), ("proj dgrad", 768, 768, "plain")]
    print("%-18s" % "us per launch" + "".join("%9s" % ("s=%d" % l) for l in levels))
    for name, n, k, kind in prods:
        sets = []
        for _ in range(nsets):
            x = torch.randn(m, k, device=dev, generator=g).bfloat16()
            w = (torch.randn(n, k, device=dev, generator=g) * 0.05).bfloat16()
            bias = torch.randn(n, device=dev, generator=g)
            y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
            aux = torch.randn(m, n, device=dev, generator=g).bfloat16()
            res = torch.randn(m, n, device=dev, generator=g) if kind == "resid" else None
            yf = torch.empty(m, n, device=dev) if kind == "resid" else None
            rs = torch.rand(m, device=dev, generator=g)
            sets.append((x, w, bias, y, aux, res, yf, rs))
        row = []
        for lv in levels:
            st = ops.EPI_STAGGER(lv)
            fns = []
            for (x, w, bias, y, aux, res, yf, rs) in sets:
                if kind == "bias":
                    fns.append(lambda x=x, w=w, y=y, bias=bias: ops.gemm(x, w, y, m, n, k, k, k, n, 0, 0, bias=bias, epilogue=st))
                elif kind == "gelu":
                    fns.append(lambda x=x, w=w, y=y, bias=bias, aux=aux: ops.gemm(
                        x, w, y, m, n, k, k, k, n, 0, 0, bias=bias, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD | st, aux=aux))
                elif kind == "resid":
                    fns.append(lambda x=x, w=w, yf=yf, bias=bias, res=res, rs=rs: ops.gemm(
                        x, w, yf, m, n, k, k, k, n, 0, 0, bias=bias, rowscale=rs, epilogue=ops.EPI_RESIDUAL | th768 | st, aux=res))
                elif kind == "gelu_bwd":
                    fns.append(lambda x=x, w=w, y=y, aux=aux: ops.gemm(
                        x, w, y, m, n, k, k, k, n, 0, 0, epilogue=ops.EPI_GELU_BWD | ops.EPI_AUX_GRAD | st, aux=aux))
                else:
                    fns.append(lambda x=x, w=w, y=y: ops.gemm(x, w, y, m, n, k, k, k, n, 0, 0,
                                                              epilogue=(th768 if n == 768 else 0) | st))
            row.append(bench(fns))
        fl = 2.0 * m * n * k
        print("%-18s" % name + "".join("%9.1f" % t for t in row) + "   best %.0f TFLOP/s (s=0: %.0f)" % (
            fl / min(row) / 1e6, fl / row[0] / 1e6))
        del sets


if __name__ == "__main__":
    main()
