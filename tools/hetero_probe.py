"""Do an HBM-bound and an MFMA-bound ROLE overlap inside ONE launch?  (round 4)

Two queues do not overlap the step's two kinds of work on this runtime (tools/overlap_probe.py, tools/cu_mask_probe.py).  Here both live
in one launch of the debug build (editor_probe_gemm_hetero): the first `nmem` workgroups stream memory (d = s0 + s1, 570 MB moved - the
bytes of a LayerNorm backward), the others run the ping-pong kernel's body on the 2 328 tiles of a 49 536 x 3 072 x 768 forward
product.  Both roles have the kernel's footprint (one workgroup per CU): the memory workgroups are dispatched first and hold nmem CUs
while the tiles cycle over the rest.
    python tools/hetero_probe.py
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editor_amd import _lib, build  # noqa: E402


def main():
    build.build(trace=True)
    tr = ctypes.CDLL(build.LIB_TRACE)
    fn = tr.editor_probe_gemm_hetero
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3 + \
                  [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    m, n, k = 3 * 128 * 129, int(os.environ.get("PROBE_N", 3072)), int(os.environ.get("PROBE_K", 768))
    print("product %d x %d x %d (%d tiles)" % (m, n, k, (m + 255) // 256 * (n // 256)))
    a = torch.randn(m, k, device=dev, generator=g).bfloat16()
    w = (torch.randn(n, k, device=dev, generator=g) * 0.05).bfloat16()
    bias = torch.randn(n, device=dev, generator=g)
    c = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    nf = 3 * 128 * 129 * 768 * 5 // 4                       # 190 MB per array: 380 MB read + 190 MB written
    s0 = torch.randn(nf, device=dev, generator=g)
    s1 = torch.randn(nf, device=dev, generator=g)
    d = torch.empty(nf, device=dev)
    n4 = nf // 4
    stream = _lib._raw_stream(torch.cuda.current_device())

    def call(with_tiles, n4_, nmem, unroll=8):
        rc = fn(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, bias.data_ptr(), with_tiles, s0.data_ptr(), s1.data_ptr(),
                d.data_ptr(), n4_, nmem, unroll, stream)
        if rc:
            raise RuntimeError("editor_probe_gemm_hetero -> %d" % rc)

    def timeit(fn_, reps=6):
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn_()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]

    # correctness of both roles in the combined launch
    call(1, n4, 32)
    torch.cuda.synchronize()
    ref = (a[:512].float() @ w.float().t() + bias).bfloat16()
    assert torch.equal(c[:512], ref) or float((c[:512].float() - ref.float()).abs().max()) < 0.1
    assert torch.equal(d, s0 + s1)
    mb = 3 * nf * 4 / 1e6
    t_g = timeit(lambda: call(1, 0, 0))
    t_m = timeit(lambda: call(0, n4, 1024))
    print("product alone on 256 CUs                         %7.1f us" % t_g)
    print("memory role alone on every CU (%d MB moved)      %7.1f us = %.2f TB/s" % (mb, t_m, mb / t_m))
    print("one after the other                              %7.1f us" % (t_g + t_m))
    for nmem in (16, 32, 48, 64, 96):
        t_mo = timeit(lambda: call(0, n4, nmem))
        t_go = timeit(lambda: call(1, 0, nmem))
        t_h = timeit(lambda: call(1, n4, nmem))
        print("nmem = %3d: memory role alone on %3d CUs %7.1f us (%.2f TB/s), product beside %3d idle workgroups %7.1f us, "
              "BOTH IN ONE LAUNCH %7.1f us  (sum of the full-chip times %.1f)" % (nmem, nmem, t_mo, mb / t_mo, nmem, t_go, t_h, t_g + t_m))
    # the REAL LayerNorm-backward role (the one editor_gemm_wgrad_group_ln carries) in place of the plain stream
    fl = tr.editor_probe_gemm_hetero_ln
    fl.restype = ctypes.c_int
    fl.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_long] + \
                  [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_void_p]
    rows, dd = 3 * 128 * 129, 768
    ldy = torch.randn(rows, dd, device=dev, generator=g).bfloat16()
    lx = torch.randn(rows, dd, device=dev, generator=g)
    gam = torch.ones(dd, device=dev)
    mean = torch.zeros(rows, device=dev)
    rstd = torch.ones(rows, device=dev)
    dxi = torch.randn(rows, dd, device=dev, generator=g)
    dxo = torch.empty(rows, dd, device=dev)
    c16 = torch.empty(rows, dd, device=dev, dtype=torch.bfloat16)
    parts = torch.empty(256 * 3 * dd, device=dev)

    def call_ln(with_tiles, nmem):
        rc = fl(a.data_ptr(), w.data_ptr(), c.data_ptr(), m, n, k, bias.data_ptr(), with_tiles, ldy.data_ptr(), lx.data_ptr(),
                gam.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, dxi.data_ptr(), dxo.data_ptr(), parts.data_ptr(),
                c16.data_ptr(), parts[256 * 2 * dd:].data_ptr(), nmem, stream)
        if rc:
            raise RuntimeError("editor_probe_gemm_hetero_ln -> %d" % rc)

    lmb = rows * dd * (2 + 4 + 4 + 4 + 2) / 1e6
    for nmem in (32, 64, 96, 256):
        t_lo = timeit(lambda: call_ln(0, nmem))
        t_lh = timeit(lambda: call_ln(1, nmem)) if nmem < 256 else float("nan")
        print("LayerNorm-backward role (%d MB), nmem = %3d: alone %7.1f us (%.1f GB/s per CU, %.2f TB/s), beside the product's tiles %7.1f us "
              "(product alone %.1f)" % (lmb, nmem, t_lo, lmb * 1e3 / t_lo / nmem, lmb / t_lo, t_lh, t_g))
    # bytes in flight per workgroup: 64 / 128 / 256 KiB
    for unroll in (4, 8, 16):
        for nmem in (32, 64):
            t_mo = timeit(lambda: call(0, n4, nmem, unroll))
            t_h = timeit(lambda: call(1, n4, nmem, unroll))
            print("%3d KiB in flight per workgroup, nmem = %2d: memory role alone %7.1f us (%.1f GB/s per CU), both in one launch %7.1f us"
                  % (unroll * 16, nmem, t_mo, mb * 1e3 / t_mo / nmem, t_h))


if __name__ == "__main__":
    main()
