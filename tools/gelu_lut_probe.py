"""One variant of the fc1 forward product (M = 3*128*129, N = 3072, K = 768) launched 8 times, for tools/pmc_lds_conflicts.sh:
    python tools/gelu_lut_probe.py none|gelu|gelu_nosave|gelu_f16
none: bias only (16-bit output, one-pass staged epilogue); gelu: + GELU with the saved gelu' (bf16: the 16 KiB LDS table of
(gelu, gelu') pairs, gathered per element); gelu_nosave: GELU, no second output (no-grad forward); gelu_f16: the f16 build
(no table: the activation in arithmetic).  VERDICT r5 item 6: is the table the 0.18 LDS-conflict ratio of <...,8,8>?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editor_amd import ops  # noqa: E402

kind = sys.argv[1]
if os.environ.get("GEMM_ALT_LIB"):          # tools/pmc_mi32.sh: the same launches out of another build of gemm_bf16.hip
    import ctypes
    from editor_amd import _lib
    _l = _lib.lib()
    _alt = ctypes.CDLL(os.environ["GEMM_ALT_LIB"])
    for _n in ("editor_gemm_bf16", "editor_gemm_f16"):
        _f = getattr(_alt, _n)
        _f.argtypes = _l.protos[_n]
        _f.restype = ctypes.c_int
        _l._fn[_n] = _f
dt = torch.float16 if kind == "gelu_f16" else torch.bfloat16
m, n, k = 3 * 128 * 129, 3072, 768
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(m, k, device="cuda", generator=g).to(dt)
w = (torch.randn(n, k, device="cuda", generator=g) * 0.04).to(dt)
bias = torch.randn(n, device="cuda", generator=g) * 0.1
c = torch.empty(m, n, dtype=dt, device="cuda")
aux = torch.empty(m, n, dtype=dt, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(9):
    if it == 1:
        e0.record()
    if kind == "none":
        ops.gemm(a, w, c, m, n, k, k, k, n, 0, 0, bias=bias)
    elif kind == "gelu_nosave":
        ops.gemm(a, w, c, m, n, k, k, k, n, 0, 0, bias=bias, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD, aux=None)
    else:
        ops.gemm(a, w, c, m, n, k, k, k, n, 0, 0, bias=bias, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD, aux=aux)
e1.record()
torch.cuda.synchronize()
print("%s: %.1f us per launch" % (kind, 1e3 * e0.elapsed_time(e1) / 8))
