"""How much does a ROUND of tiles cost beyond the tile itself?  t(r rounds) for r = 1 .. 4 (M = r x 21 760 rows, N = 768 or 2 304,
K = 768, 256-row tiles, call by call with idle gaps): the slope is the time per round, to be held against the in-kernel tile time of
tools/gemm_tile_ends.py - the difference is what the hardware needs to retire 256 workgroups and start the next 256.
    python tools/gemm_round_gap.py [--persist]      (--persist: EDITOR_EPI_PERSIST, one workgroup per CU walking its tiles)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editor_amd import ops  # noqa: E402
from tools.gemm_bound_probe import timed  # noqa: E402


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    gap = lambda: torch.cuda._sleep(400000)
    flags = [("plain launch", ops.EPI_FORCE_PP)]
    if hasattr(ops, "EPI_PERSIST"):
        flags.append(("persistent", ops.EPI_FORCE_PP | ops.EPI_PERSIST))
    for n, k in ((768, 768), (2304, 768), (768, 3072)):
        per_round_rows = 256 * (255 // (n // 256))
        for name, fl in flags:
            ts = []
            for r in (1, 2, 3, 4):
                m = per_round_rows * r
                x = torch.randn(m, k, device=dev, generator=g).bfloat16()
                w = (torch.randn(n, k, device=dev, generator=g) * 0.05).bfloat16()
                y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
                t, _ = timed(lambda: ops.gemm(x, w, y, m, n, k, k, k, n, 0, 0, epilogue=fl), gap)
                ts.append(t)
            slope = (ts[3] - ts[0]) / 3.0
            print("N=%-5d K=%-5d %-12s 1..4 rounds: %s us   -> %.1f us per additional round"
                  % (n, k, name, " ".join("%6.1f" % t for t in ts), slope), flush=True)


if __name__ == "__main__":
    main()
