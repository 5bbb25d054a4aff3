"""Do an MFMA-bound and an HBM-bound kernel of the step overlap when they come from two streams?  (Idea: run the backbone as
two half-batches on two streams, one block phase apart, so that one half's LayerNorm / attention / epilogue traffic runs
under the other half's GEMM main loops.)  Serial = both chains on one stream; concurrent = one chain per stream.
    python tools/overlap_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editor_amd import ops  # noqa: E402


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    for m in (3 * 128 * 129, 3 * 64 * 129):
        d, hid = 768, 3072
        x = torch.randn(m, d, device=dev, generator=g)
        h = torch.randn(m, d, device=dev, generator=g).bfloat16()
        w1 = (torch.randn(hid, d, device=dev, generator=g) * 0.05).bfloat16()
        w2 = (torch.randn(d, hid, device=dev, generator=g) * 0.05).bfloat16()
        y1 = torch.empty(m, hid, device=dev, dtype=torch.bfloat16)
        aux = torch.empty(m, hid, device=dev, dtype=torch.bfloat16)
        bias1 = torch.randn(hid, device=dev, generator=g)
        gam, bet = torch.ones(d, device=dev), torch.zeros(d, device=dev)
        x2 = torch.randn(m, d, device=dev, generator=g)
        res = torch.randn(m, d, device=dev, generator=g)
        yo = torch.empty(m, d, device=dev)
        bias2 = torch.randn(d, device=dev, generator=g)
        n_it = 6

        def gemms():
            for _ in range(n_it):
                ops.gemm(h, w1, y1, m, hid, d, d, d, hid, 0, 0, bias=bias1, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD, aux=aux)
                ops.gemm(y1, w2, yo, m, d, hid, hid, hid, d, 0, 0, bias=bias2, epilogue=ops.EPI_RESIDUAL, aux=res)

        def mems():
            for _ in range(n_it * 4):
                ops.layernorm_fwd(x2, gam, bet, 1e-6, torch.bfloat16)

        def serial():
            gemms()
            mems()

        def concurrent(sa, sb):
            cur = torch.cuda.current_stream()
            sa.wait_stream(cur)
            sb.wait_stream(cur)
            with torch.cuda.stream(sa):
                mems()
            with torch.cuda.stream(sb):
                gemms()
            cur.wait_stream(sa)
            cur.wait_stream(sb)
        tg, tm = timeit(gemms), timeit(mems)
        ts = timeit(serial)
        print("M = %6d: GEMM chain %.2f ms, LayerNorm chain %.2f ms, one stream %.2f ms  (sum %.2f, max %.2f)"
              % (m, tg, tm, ts, tg + tm, max(tg, tm)), flush=True)
        for name, pa, pb in (("equal priority", 0, 0), ("GEMM stream high", 0, -1), ("LayerNorm stream high", -1, 0)):
            sa, sb = torch.cuda.Stream(priority=pa), torch.cuda.Stream(priority=pb)
            print("    two streams, %-22s %.2f ms" % (name, timeit(lambda: concurrent(sa, sb))), flush=True)


if __name__ == "__main__":
    main()
