"""Bring-up of the four-wave GEMM tile (csrc/gemm_w4.hip): bit-compare with the ping-pong kernel, then time both per K-tile with
the method of tools/gemm_bound_probe.py (one round of tiles, call by call, warm / cold operands, K = 768 vs 3072 by difference).
    python tools/gemm_w4_probe.py
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editor_amd import _lib, ops  # noqa: E402
from tools.gemm_bound_probe import timed  # noqa: E402


def w4(x, w, y, m, n, k, f16=0, ablate=0):
    fn = _lib.probe_lib().editor_probe_gemm_w4
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_long] * 3 + [ctypes.c_int, ctypes.c_void_p]
    rc = fn(x.data_ptr(), w.data_ptr(), y.data_ptr(), f16, m, n, k, k, k, n, ablate, torch.cuda.current_stream().cuda_stream)
    if rc:
        raise RuntimeError("editor_probe_gemm_w4 -> %d" % rc)


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    for (m, n, k) in ((512, 256, 64), (512, 256, 128), (777, 768, 768), (21760, 768, 3072), (5376, 3072, 768)):
        x = torch.randn(m, k, device=dev, generator=g).bfloat16()
        w = (torch.randn(n, k, device=dev, generator=g) * 0.05).bfloat16()
        y0 = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        y1 = torch.full((m, n), 7.0, device=dev, dtype=torch.bfloat16)
        ops.gemm(x, w, y0, m, n, k, k, k, n, 0, 0, epilogue=ops.EPI_FORCE_PP if m >= 256 else 0)
        w4(x, w, y1, m, n, k)
        torch.cuda.synchronize()
        ref = (x.float() @ w.float().t())
        print("M=%-6d N=%-5d K=%-5d  == ping-pong kernel: %s   max |err| vs fp32 matmul: %.3e (ping-pong %.3e)"
              % (m, n, k, torch.equal(y0, y1), (y1.float() - ref).abs().max().item(), (y0.float() - ref).abs().max().item()))
    big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    gaps = {"warm": lambda: torch.cuda._sleep(400000), "cold": lambda: big.fill_(1)}
    res = {}
    for n in (768, 3072):
        m = 256 * (255 // (n // 256))
        for k in (768, 3072):
            x = torch.randn(m, k, device=dev, generator=g).bfloat16()
            w = (torch.randn(n, k, device=dev, generator=g) * 0.05).bfloat16()
            y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
            for name, gap in gaps.items():
                res[("pp", n, k, name)], _ = timed(lambda: ops.gemm(x, w, y, m, n, k, k, k, n, 0, 0, epilogue=ops.EPI_FORCE_PP), gap)
                res[("w4", n, k, name)], _ = timed(lambda: w4(x, w, y, m, n, k), gap)
                print("N=%-5d K=%-5d %-4s  ping-pong %6.1f us   four-wave %6.1f us" % (n, k, name, res[("pp", n, k, name)],
                                                                                      res[("w4", n, k, name)]))
    for n in (768, 3072):
        for name in gaps:
            for kern in ("pp", "w4"):
                per = (res[(kern, n, 3072, name)] - res[(kern, n, 768, name)]) / 36.0
                print("N=%-5d %-4s %s: %.3f us per K-tile (the matrix core alone: 0.859), tile ends %.1f us"
                      % (n, name, kern, per, res[(kern, n, 768, name)] - 12 * per))


def ablations():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    n, m = 768, 256 * 85
    gap = lambda: torch.cuda._sleep(400000)
    ops_ = {}
    for k in (768, 3072):
        ops_[k] = (torch.randn(m, k, device=dev, generator=g).bfloat16(), (torch.randn(n, k, device=dev, generator=g) * 0.05).bfloat16(),
                   torch.empty(m, n, device=dev, dtype=torch.bfloat16))
    for abl, what in ((0, "complete"), (1, "no LDS-DMA in the loop"), (4, "no fragment reads"), (5, "MFMAs + barriers"),
                      (2, "no MFMAs"), (6, "LDS-DMA + barriers"), (3, "reads + barriers"), (7, "barriers only")):
        t = {}
        for k in (768, 3072):
            x, w, y = ops_[k]
            t[k], _ = timed(lambda: w4(x, w, y, m, n, k, 0, abl), gap)
        per = (t[3072] - t[768]) / 36.0
        print("four-wave, warm, ablate %d %-24s K=768 %6.1f us  K=3072 %6.1f us -> %.3f us per K-tile" % (abl, what, t[768], t[3072], per))


if __name__ == "__main__":
    if "--ablate" in sys.argv:
        ablations()
        sys.exit(0)
    main()
