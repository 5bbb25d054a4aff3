"""What clock do the CUs run at while the GEMMs of the step keep the matrix cores busy?  One resident wave samples
(s_memtime = shader-clock cycles, s_memrealtime = 100 MHz reference) while another stream runs a chosen product back to back.
    python tools/clock_probe.py          # idle, then under each of the path's forward products (bf16, M = 49 536)
MI355X_MICROARCH.md prices the MFMA peak (2.5 PFLOP/s bf16 dense) at the 2.4 GHz boost clock."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editor_amd import _lib, ops  # noqa: E402


def trace(work, n=2400, sleep=6, label="", series=False):
    probe = _lib.probe_lib()
    buf = torch.zeros(2 * n, dtype=torch.int64, device="cuda")
    s_probe, s_work = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    fn = probe.editor_probe_clock_trace
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    fn.restype = ctypes.c_int
    rc = fn(buf.data_ptr(), n, sleep, s_probe.cuda_stream)
    assert rc == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s_work):
        e0.record()
        if work is not None:
            for _ in range(work[1]):
                work[0]()
        e1.record()
    torch.cuda.synchronize()
    t = buf.view(n, 2).cpu().numpy().astype("float64")
    cyc, ref = t[:, 0] - t[0, 0], (t[:, 1] - t[0, 1]) / 100.0          # shader cycles, microseconds
    span = ref[-1]
    busy = e0.elapsed_time(e1) * 1e3
    # frequency over the middle of the busy window (or the whole trace when idle)
    lo, hi = (0.25 * min(busy, span), 0.75 * min(busy, span)) if work is not None else (0.1 * span, 0.9 * span)
    i0, i1 = (ref >= lo).argmax(), (ref >= hi).argmax()
    mhz = (cyc[i1] - cyc[i0]) / (ref[i1] - ref[i0])
    print("%-28s shader clock %7.0f MHz   (trace %.1f ms, work %.1f ms)" % (label, mhz, span / 1e3, busy / 1e3), flush=True)
    if series:                                         # MHz per 5 ms window: how fast the power management settles
        pts = []
        for t0 in range(0, int(min(busy, span) / 1e3) - 4, 5):
            j0, j1 = (ref >= t0 * 1e3).argmax(), (ref >= (t0 + 5) * 1e3).argmax()
            if j1 > j0:
                pts.append("%d" % round((cyc[j1] - cyc[j0]) / (ref[j1] - ref[j0])))
        print("    MHz per 5 ms: " + " ".join(pts), flush=True)
    return mhz


def mfma_peak(zero):
    probe = _lib.probe_lib()
    fn = probe.editor_probe_mfma_peak
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    fn.restype = ctypes.c_int
    out = torch.zeros(4, device="cuda")
    grid, iters = 256 * 2, 40000                        # 8 waves per CU; ~1 ms per launch at full rate
    fl = grid * 4 * iters * 8 * 16384.0

    def f():
        assert fn(out.data_ptr(), grid, iters, zero, torch.cuda.current_stream().cuda_stream) == 0
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(60):
        f()
    e1.record()
    torch.cuda.synchronize()
    per = e0.elapsed_time(e1) / 60
    mhz = trace((f, 60), label="MFMA only (%s operands)" % ("zero" if zero else "random"), series=True)
    print("    %.1f TFLOP/s sustained = %.3f of 2 500; at the measured clock the issue-rate ceiling is %.0f"
          % (fl / per / 1e9, fl / per / 1e9 / 2500, 2500 * mhz / 2400), flush=True)


def main():
    m = 3 * 128 * 129
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    trace(None, label="idle")
    mfma_peak(0)
    mfma_peak(1)
    for (n, k) in ((2304, 768), (3072, 768), (768, 3072), (768, 768)):
        x = torch.randn(m, k, device=dev, generator=g).bfloat16()
        w = (torch.randn(n, k, device=dev, generator=g) * 0.05).bfloat16()
        y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        fl = 2.0 * m * n * k
        def f():
            ops.gemm(x, w, y, m, n, k, k, k, n, 0, 0)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            f()
        e1.record()
        torch.cuda.synchronize()
        per = e0.elapsed_time(e1) / 50
        reps = max(20, int(45.0 / per))
        mhz = trace((f, reps), label="bf16 fwd N=%d K=%d" % (n, k), series=True)
        print("    %.1f TFLOP/s = %.3f of the 2.4 GHz peak, %.3f of the peak at the measured clock"
              % (fl / per / 1e9, fl / per / 1e9 / 2500, fl / per / 1e9 / (2500 * mhz / 2400)), flush=True)
    # elementwise / HBM-bound work for comparison
    a = torch.randn(64 * 1024 * 1024, device=dev)
    trace((lambda: a.mul_(1.0001), 200), label="fp32 elementwise (HBM)")


if __name__ == "__main__":
    main()
