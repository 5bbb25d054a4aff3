"""Accuracy of the 16-bit attention kernel variants against a float64 reference on the same (rounded) inputs:
dense body+tail kernels (T = 16n+1), dense whole-sequence kernels (FULL paths), packed variable-length kernels."""
import sys, torch
sys.path.insert(0, '.')
from editor_amd import ops

def ref(qkv, do, b, t, heads, hd):
    d = heads * hd
    q = qkv.double().requires_grad_(True)
    qq, kk, vv = (q[:, i * d:(i + 1) * d].reshape(b, t, heads, hd).transpose(1, 2) for i in range(3))
    p = ((qq @ kk.transpose(-2, -1)) * hd ** -0.5).softmax(-1)
    o = (p @ vv).transpose(1, 2).reshape(b * t, d)
    o.backward(do.double())
    return o.detach(), q.grad

def rel(a, b):
    return ((a.double() - b).norm() / b.norm()).item()

for dtype in (torch.float16, torch.bfloat16):
    for t in (129, 130, 193, 513):
        b, heads, hd = 8, 12, 64
        g = torch.Generator().manual_seed(t)
        qkv = (torch.randn(b * t, 3 * heads * hd, generator=g) * 0.8).to(dtype)
        do = (torch.randn(b * t, heads * hd, generator=g) * 1e-3).to(dtype)
        o_ref, dq_ref = ref(qkv.cpu(), do.cpu(), b, t, heads, hd)
        qg, dg = qkv.cuda(), do.cuda()
        o, lse = ops.attention_fwd(qg, b, t, heads, hd)
        dq = ops.attention_bwd(qg, dg, b, t, heads, hd, None, lse, o)
        cu = (torch.arange(b + 1, dtype=torch.int32) * t).cuda()
        o2, lse2 = ops.attention_fwd(qg, b, t, heads, hd, cu=cu)
        dq2 = ops.attention_bwd(qg, dg, b, t, heads, hd, None, lse2, o2, cu=cu)
        d = heads * hd
        parts = lambda x: [rel(x.cpu()[:, i * d:(i + 1) * d], dq_ref[:, i * d:(i + 1) * d]) for i in range(3)]
        print("%s T=%d  dense: o %.2e dq/dk/dv %s | varlen: o %.2e dq/dk/dv %s | lse diff %.1e" % (
            str(dtype)[6:], t, rel(o.cpu(), o_ref), ["%.2e" % v for v in parts(dq)], rel(o2.cpu(), o_ref),
            ["%.2e" % v for v in parts(dq2)], (lse - lse2).abs().max().item()))
