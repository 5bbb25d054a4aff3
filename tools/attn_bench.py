import torch, sys
sys.path.insert(0, '.')
from editor_amd import ops
b, t, heads, hd = 384, 129, 12, 64
qkv = (torch.randn(b * t, 3 * heads * hd, device='cuda') * 0.5).bfloat16()
ldp = (t + 3) // 4 * 4
probs = torch.empty(b, heads, t, ldp, device='cuda')
def bench(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
print("fwd with probs  %.1f us" % bench(lambda: ops.attention_fwd(qkv, b, t, heads, hd, None, probs)))
print("fwd no probs    %.1f us" % bench(lambda: ops.attention_fwd(qkv, b, t, heads, hd, None, None)))

o, lse = ops.attention_fwd(qkv, b, t, heads, hd, None, None)
do = torch.randn_like(o)
print("bwd (dq + dk/dv)  %.1f us" % bench(lambda: ops.attention_bwd(qkv, do, b, t, heads, hd, None, lse, o)))
print("rollout step      %.1f us" % bench(lambda: ops.attn_rollout_qk([(qkv, lse)], b, t, heads, hd)))
for tt in (193,):
    q2 = (torch.randn(b * tt, 3 * heads * hd, device='cuda') * 0.5).bfloat16()
    print("T=%d fwd no probs %.1f us" % (tt, bench(lambda: ops.attention_fwd(q2, b, tt, heads, hd, None, None))))
for mode, name in ((0, "two-pass"), (1, "fused")):
    ops.attention_bwd_mode(mode)
    print("bwd %-8s        %.1f us" % (name, bench(lambda: ops.attention_bwd(qkv, do, b, t, heads, hd, None, lse, o))))
ops.attention_bwd_mode(0)
