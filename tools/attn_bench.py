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
# (round 6: the first step of a rollout - the one-hot class-token vector - reads one query tile; a later step all of them: two layers minus one)
r1 = bench(lambda: ops.attn_rollout_qk([(qkv, lse)], b, t, heads, hd))
r2 = bench(lambda: ops.attn_rollout_qk([(qkv, lse), (qkv, lse)], b, t, heads, hd))
print("rollout step      %.1f us   (first, one-hot step: %.1f us)" % (r2 - r1, r1))
for tt in (193,):
    q2 = (torch.randn(b * tt, 3 * heads * hd, device='cuda') * 0.5).bfloat16()
    print("T=%d fwd no probs %.1f us" % (tt, bench(lambda: ops.attention_fwd(q2, b, tt, heads, hd, None, None))))
# round 4: the fused kernels at the factory's other head widths (same 768 = heads x hd columns, B = 384 sequences of 129 tokens)
# against the exact-f32 kernels between two casts they used to take (ops.attention_fwd's fallback for other widths)
for heads2, hd2 in ((8, 96), (24, 32)):
    q3 = (torch.randn(b * t, 3 * heads2 * hd2, device='cuda') * 0.5).bfloat16()
    o3, lse3 = ops.attention_fwd(q3, b, t, heads2, hd2, None, None)
    do3 = torch.randn_like(o3)
    f16 = bench(lambda: ops.attention_fwd(q3, b, t, heads2, hd2, None, None))
    b16 = bench(lambda: ops.attention_bwd(q3, do3, b, t, heads2, hd2, None, lse3, o3))
    r16 = (bench(lambda: ops.attn_rollout_qk([(q3, lse3), (q3, lse3)], b, t, heads2, hd2))
           - bench(lambda: ops.attn_rollout_qk([(q3, lse3)], b, t, heads2, hd2)))
    q32 = q3.float()
    def detour_fwd():
        o_, p_ = ops.attention_fwd(q3.float(), b, t, heads2, hd2, None, None)
        return o_.to(q3.dtype), p_
    _, p32 = detour_fwd()
    def detour_bwd():
        return ops.attention_bwd(q3.float(), do3.float(), b, t, heads2, hd2, None, p32, None).to(q3.dtype)
    print("hd=%d (%d heads): fwd %.1f us, bwd %.1f us, rollout step %.1f us   |  exact-f32 detour: fwd %.1f us, bwd %.1f us"
          % (hd2, heads2, f16, b16, r16, bench(detour_fwd, 5), bench(detour_bwd, 5)))
