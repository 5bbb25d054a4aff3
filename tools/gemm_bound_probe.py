"""What bounds the ping-pong GEMM's main loop in the step - the matrix core or operand delivery?  (round 4)

One ROUND of 256 x 256 tiles (<= 256 workgroups, so the launch time IS one tile's time: prologue + K/64 K-tiles + epilogue) of the
hot path's two extreme shapes, timed call by call with HIP events under the step's conditions rather than back to back:
  warm: the same operands every call, an idle gap between calls (boost clock, A resident in the 256 MB Infinity Cache)
  cold: a 512 MB fill between calls (boost clock, A comes from HBM - what a block's GEMMs see in the step)
and for two reduction lengths, so that the K-tile cost comes out by difference (the tile ends cancel):
    us per K-tile = (t[K = 3072] - t[K = 768]) / 36
Run it again under a capped shader clock (tools/gemm_bound_probe.sh tries `rocm-smi --setperfdeterminism`): a matrix-core-bound loop
scales with the clock, a delivery-bound one does not.
    python tools/gemm_bound_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editor_amd import ops  # noqa: E402


def timed(fn, gap, iters=24):
    ts = []
    for i in range(iters + 4):
        gap()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if i >= 4:
            ts.append(e0.elapsed_time(e1) * 1000.0)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    gaps = {"warm": lambda: torch.cuda._sleep(400000), "cold": lambda: big.fill_(1)}
    res = {}
    for n in (768, 3072):
        m = 256 * (255 // (n // 256))                       # one round: tiles_m * tiles_n <= 256
        for k in (768, 3072):
            x = torch.randn(m, k, device=dev, generator=g).bfloat16()
            w = (torch.randn(n, k, device=dev, generator=g) * 0.05).bfloat16()
            y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
            tiles = (m // 256) * (n // 256)
            for name, gap in gaps.items():
                med, best = timed(lambda: ops.gemm(x, w, y, m, n, k, k, k, n, 0, 0, epilogue=ops.EPI_FORCE_PP), gap)
                res[(n, k, name)] = med
                print("N=%-5d K=%-5d M=%-6d (%3d tiles)  %-4s  median %7.1f us  best %7.1f us   %7.1f TFLOP/s"
                      % (n, k, m, tiles, name, med, best, 2.0 * m * n * k / med / 1e6))
    for n in (768, 3072):
        for name in gaps:
            per = (res[(n, 3072, name)] - res[(n, 768, name)]) / 36.0
            ends = res[(n, 768, name)] - 12 * per
            print("N=%-5d %-4s: %.3f us per K-tile (%.0f cycles at 2.4 GHz; the matrix core alone: 0.859 us), tile ends %.1f us"
                  % (n, name, per, per * 2400, ends))
    if "--ablate" in sys.argv:
        # by deletion, in the DEBUG build of the GEMM (libeditor_gemm_trace.so; EDITOR_GEMM_ABLATE: 1 = no LDS-DMA inside the K loop,
        # 2 = no MFMAs, 4 = no fragment reads): what the loop costs without each of its three activities (results are garbage)
        from tools.gemm_bench import use_trace_build
        use_trace_build()
        n = 768
        m = 256 * (255 // (n // 256))
        ops_ = {}
        for k in (768, 3072):
            x = torch.randn(m, k, device=dev, generator=g).bfloat16()
            w = (torch.randn(n, k, device=dev, generator=g) * 0.05).bfloat16()
            y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
            ops_[k] = (x, w, y)
        for abl, what in ((0, "complete"), (1, "no LDS-DMA in the loop"), (2, "no MFMAs"), (4, "no fragment reads"),
                          (3, "no LDS-DMA, no MFMAs (reads + barriers)"), (6, "no MFMAs, no reads (LDS-DMA + barriers)"),
                          (5, "no LDS-DMA, no reads (MFMAs + barriers)"), (7, "barriers only")):
            os.environ["EDITOR_GEMM_ABLATE"] = str(abl)
            for name, gap in gaps.items():
                t = {}
                for k in (768, 3072):
                    x, w, y = ops_[k]
                    t[k], _ = timed(lambda: ops.gemm(x, w, y, m, n, k, k, k, n, 0, 0, epilogue=ops.EPI_FORCE_PP), gap)
                per = (t[3072] - t[768]) / 36.0
                print("ablate %d %-4s %-44s K=768 %6.1f us  K=3072 %6.1f us  -> %.3f us per K-tile (%4.0f cycles at 2.4 GHz)"
                      % (abl, name, what, t[768], t[3072], per, per * 2400))
        os.environ["EDITOR_GEMM_ABLATE"] = "0"


if __name__ == "__main__":
    main()
