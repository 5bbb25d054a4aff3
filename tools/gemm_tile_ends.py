"""Per-workgroup phase timeline (debug build, EDITOR_GEMM_TRACE=1) of ONE ROUND of tiles at the step's conditions (idle gap between
calls: boost clock, operands warm): how long prologue, K loop, the two epilogue stages and the store drain of a tile take, per
epilogue kind.    python tools/gemm_tile_ends.py 2> gpurun_out/gemm_tile_ends.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["EDITOR_GEMM_TRACE"] = "1"
from editor_amd import ops  # noqa: E402
from tools.gemm_bench import use_trace_build  # noqa: E402


def main():
    use_trace_build()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    for (n, k) in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
        m = 256 * (255 // (n // 256))
        x = torch.randn(m, k, device=dev, generator=g).bfloat16()
        w = (torch.randn(n, k, device=dev, generator=g) * 0.05).bfloat16()
        bias = torch.randn(n, device=dev, generator=g)
        y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        yf = torch.empty(m, n, device=dev)
        pre = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        res = torch.randn(m, n, device=dev, generator=g)
        rs = torch.rand(m, device=dev, generator=g)
        kinds = (("bias, 16-bit out", lambda: ops.gemm(x, w, y, m, n, k, k, k, n, 0, 0, bias=bias, epilogue=ops.EPI_FORCE_PP)),
                 ("bias + GELU, two outputs", lambda: ops.gemm(x, w, y, m, n, k, k, k, n, 0, 0, bias=bias,
                                                            epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD | ops.EPI_FORCE_PP, aux=pre)),
                 ("bias + fp32 residual", lambda: ops.gemm(x, w, yf, m, n, k, k, k, n, 0, 0, bias=bias, rowscale=rs,
                                                        epilogue=ops.EPI_RESIDUAL | ops.EPI_FORCE_PP, aux=res)))
        for name, fn in kinds:
            sys.stderr.write("--- N=%d K=%d M=%d  %s\n" % (n, k, m, name))
            for _ in range(3):
                torch.cuda._sleep(400000)
                fn()
                torch.cuda.synchronize()


if __name__ == "__main__":
    main()
