"""Can an MFMA-bound and an HBM-bound chain of the step run SIDE BY SIDE on disjoint sets of CUs?  (round 4)

tools/overlap_probe.py found that two streams do not overlap the two kinds of work: the GEMM's workgroups fill every CU, the
LayerNorm's wait for them.  Here each stream is created with a CU MASK (hipExtStreamCreateWithCUMask): the GEMM chain on a fraction
f of the CUs, the memory-bound chain on the rest - if the memory-bound kernels keep most of their HBM rate on fewer CUs, the pair
finishes in ~max(t_gemm / f, t_mem') instead of t_gemm + t_mem.
    python tools/cu_mask_probe.py
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editor_amd import ops  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    """bits: iterable of 256 0/1 flags (CU i enabled) -> torch ExternalStream"""
    words = (ctypes.c_uint32 * 8)()
    for i, b in enumerate(bits):
        if b:
            words[i // 32] |= (1 << (i % 32))
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask -> %d" % rc)
    return torch.cuda.ExternalStream(s.value)


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
    m, d, hid = 3 * 128 * 129, 768, 3072
    h = torch.randn(m, d, device=dev, generator=g).bfloat16()
    dy = torch.randn(m, hid, device=dev, generator=g).bfloat16()
    dw = torch.empty(hid, d, device=dev)
    w1 = (torch.randn(hid, d, device=dev, generator=g) * 0.05).bfloat16()
    y1 = torch.empty(m, hid, device=dev, dtype=torch.bfloat16)
    x2 = torch.randn(m, d, device=dev, generator=g)
    gam, bet = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    n_it = 6

    def gemms():                       # the weight-gradient-like chain: long reductions, little HBM traffic
        for _ in range(n_it):
            ops.gemm(dy, h, dw, hid, d, m, hid, d, d, 1, 1, splitk=7, epilogue=ops.EPI_FORCE_PP)

    def fwd_gemms():                   # a forward product (tile ends with HBM traffic of their own)
        for _ in range(n_it):
            ops.gemm(h, w1, y1, m, hid, d, d, d, hid, 0, 0)

    def mems():
        for _ in range(n_it * 5):
            ops.layernorm_fwd(x2, gam, bet, 1e-6, torch.bfloat16)

    def run(chain_a, chain_b, sa, sb):
        cur = torch.cuda.current_stream()
        sa.wait_stream(cur)
        sb.wait_stream(cur)
        with torch.cuda.stream(sa):
            chain_a()
        with torch.cuda.stream(sb):
            chain_b()
        cur.wait_stream(sa)
        cur.wait_stream(sb)

    for gname, gfn in (("weight-gradient chain", gemms), ("forward-product chain", fwd_gemms)):
        tg, tm = timeit(gfn), timeit(mems)
        print("%s %.2f ms, LayerNorm chain %.2f ms: one after the other %.2f ms" % (gname, tg, tm, tg + tm), flush=True)
        plain_a, plain_b = torch.cuda.Stream(), torch.cuda.Stream()
        print("    two plain streams                                   %.2f ms" % timeit(lambda: run(gfn, mems, plain_a, plain_b)), flush=True)
        for pattern in ("interleaved", "blocked"):
            for f8 in (7, 6, 5, 4):                          # GEMM on f8 / 8 of the CUs
                if pattern == "interleaved":                 # CU i -> slot i % 8
                    ga = [1 if (i % 8) < f8 else 0 for i in range(256)]
                else:                                        # CU i -> slot (i // 4) % 8 (blocks of four)
                    ga = [1 if ((i // 4) % 8) < f8 else 0 for i in range(256)]
                gb = [1 - b for b in ga]
                sa, sb = masked_stream(ga), masked_stream(gb)
                t_g_alone = timeit(lambda: run(gfn, lambda: None, sa, sb))
                t_m_alone = timeit(lambda: run(lambda: None, mems, sa, sb))
                t_both = timeit(lambda: run(gfn, mems, sa, sb))
                print("    masks %-11s GEMM on %d/8 of the CUs: GEMM alone %.2f ms, LayerNorm alone (on %d/8) %.2f ms, together %.2f ms"
                      % (pattern, f8, t_g_alone, 8 - f8, t_m_alone, t_both), flush=True)


if __name__ == "__main__":
    main()
