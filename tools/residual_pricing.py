"""VERDICT r3 item 2, "price the bf16-branch residual with a measurement": the projection / fc2 products with their fp32 residual
epilogue + the LayerNorm that follows (today's path) against the same products with the PLAIN 16-bit epilogue + a fused
residual-add + LayerNorm (libeditor_probe.so: editor_probe_resid_add_layernorm), at the bench workload's M = 49 536 token rows,
HIP events over rotating operand sets (> 512 MB: HBM, not the Infinity Cache).      python tools/residual_pricing.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editor_amd import _lib, ops  # noqa: E402

M, D, H = 3 * 128 * 129, 768, 3072
dev = torch.device("cuda")
NS = 4
fn = _lib.probe_lib().editor_probe_resid_add_layernorm
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_float, ctypes.c_long, ctypes.c_int] + [ctypes.c_void_p] * 5


def ev(f, reps=24):
    for i in range(NS):
        f(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        f(i % NS)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


g, be = torch.ones(D, device=dev), torch.zeros(D, device=dev)
rs = (torch.rand(M, device=dev) > 0.1).float() / 0.9
for name, K in (("proj (K = 768)", D), ("fc2 (K = 3072)", H)):
    a = [(torch.randn(M, K, device=dev) * 0.5).bfloat16() for _ in range(NS)]
    w = (torch.randn(D, K, device=dev) * 0.02).bfloat16()
    bias = torch.zeros(D, device=dev)
    x = [torch.randn(M, D, device=dev) for _ in range(NS)]
    x1 = [torch.empty(M, D, device=dev) for _ in range(NS)]
    br = [torch.empty(M, D, dtype=torch.bfloat16, device=dev) for _ in range(NS)]
    y = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def gemm_resid(i):
        ops.gemm(a[i], w, x1[i], M, D, K, K, K, D, 0, 0, bias=bias, rowscale=rs, epilogue=ops.EPI_RESIDUAL, aux=x[i])

    def ln(i):
        ops.layernorm_fwd(x1[i], g, be, 1e-6, torch.bfloat16)

    def gemm_plain(i):
        ops.gemm(a[i], w, br[i], M, D, K, K, K, D, 0, 0, bias=bias)

    def add_ln(i):
        rc = fn(x[i].data_ptr(), br[i].data_ptr(), rs.data_ptr(), g.data_ptr(), be.data_ptr(), 1e-6, M, D, x1[i].data_ptr(),
                y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), st)
        assert rc == 0

    t = {k: ev(f) for k, f in (("gemm + fp32 residual epilogue", gemm_resid), ("layernorm_fwd", ln),
                               ("gemm, plain 16-bit epilogue", gemm_plain), ("fused residual-add + layernorm", add_ln))}
    today = ev(lambda i: (gemm_resid(i), ln(i)))
    alt = ev(lambda i: (gemm_plain(i), add_ln(i)))
    # the alternative rounds the branch output to bf16 before the add: how far that moves x1
    gemm_resid(0); gemm_plain(0); add_ln(0)
    ref = x1[0].clone(); gemm_resid(0)
    err = ((x1[0] - ref).norm() / (x1[0] - x[0]).norm()).item()
    print("%-16s %s" % (name, "  ".join("%s %.1f us" % kv for kv in t.items())))
    print("%-16s today (epilogue + LN) %.1f us   bf16 branch + fused add-LN %.1f us   -> %+.1f us per pair; branch rounded to bf16: "
          "rel. change of the branch term %.1e" % ("", today, alt, alt - today, err))
