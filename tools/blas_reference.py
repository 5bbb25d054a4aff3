"""A yardstick, not a dependency: the vendor BLAS (hipBLASLt / rocBLAS through torch.nn.functional.linear and `dy.t() @ x`)
on the hot path's GEMM shapes (M = 49 536 token rows, bf16), beside this repo's kernel on the same operands.  Nothing on the
product path calls a BLAS library.   python tools/blas_reference.py"""
import torch, sys
sys.path.insert(0, '.')
from editor_amd import ops
m = 3 * 128 * 129
def bench(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
g = torch.Generator(device='cuda').manual_seed(0)
for (n, k) in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    x = torch.randn(m, k, device='cuda', generator=g).bfloat16()
    w = (torch.randn(n, k, device='cuda', generator=g) * 0.05).bfloat16()
    b = torch.randn(n, device='cuda', generator=g).bfloat16()
    y = torch.empty(m, n, device='cuda', dtype=torch.bfloat16)
    fl = 2.0 * m * n * k
    t_lib = bench(lambda: torch.nn.functional.linear(x, w, b))
    bias = b.float()
    t_own = bench(lambda: ops.gemm(x, w, y, m, n, k, k, k, n, 0, 0, bias=bias))
    # wgrad-shaped: dW = dy^T x
    dy = torch.randn(m, n, device='cuda', generator=g).bfloat16()
    t_lib_w = bench(lambda: dy.t() @ x)
    print("N=%-5d K=%-5d  fwd+bias: vendor BLAS (torch F.linear) %7.1f TFLOP/s   this repo %7.1f   | wgrad-shaped vendor %7.1f" % (n, k, fl / t_lib / 1e9, fl / t_own / 1e9, fl / t_lib_w / 1e9), flush=True)

# the attention core beside torch's scaled_dot_product_attention (whatever backend this build selects), same sizes
import torch.nn.functional as F
b, t, heads, hd = 384, 129, 12, 64
qkv = (torch.randn(b * t, 3 * heads * hd, device='cuda', generator=g) * 0.5).bfloat16()
q, k, v = (qkv.view(b, t, 3, heads, hd)[:, :, i].transpose(1, 2).contiguous().requires_grad_(True) for i in range(3))
do = torch.randn(b, heads, t, hd, device='cuda', generator=g).bfloat16()
try:
    t_f = bench(lambda: F.scaled_dot_product_attention(q, k, v))
    o = F.scaled_dot_product_attention(q, k, v)
    def fb():
        o_ = F.scaled_dot_product_attention(q, k, v)
        o_.backward(do)
    t_fb = bench(fb)
    lib = "torch SDPA fwd %.1f us, fwd+bwd %.1f us" % (t_f * 1e3, t_fb * 1e3)
except Exception as e:                                     # no fused backend in this build
    lib = "torch SDPA unavailable (%s)" % type(e).__name__
o2, lse = ops.attention_fwd(qkv, b, t, heads, hd, None, None)
do2 = torch.randn_like(o2)
t_of = bench(lambda: ops.attention_fwd(qkv, b, t, heads, hd, None, None))
t_ob = bench(lambda: ops.attention_bwd(qkv, do2, b, t, heads, hd, None, lse, o2))
print("attention, 384 x 12 heads x 129 tokens x 64: %s   | this repo fwd %.1f us, bwd %.1f us (packed qkv in, no transposes)"
      % (lib, t_of * 1e3, t_ob * 1e3), flush=True)
