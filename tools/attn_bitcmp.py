"""Experiment: bitwise comparison of two builds of the attention kernels on identical inputs (run twice with
EDITOR_LIB_VARIANT set / unset, dumps to gpurun_out/attn_<tag>.pt)."""
import os, sys, torch
sys.path.insert(0, '.')
from editor_amd import _lib
v = os.environ.get("EDITOR_LIB_VARIANT")
if v:
    _lib.LIB_PATH = os.path.abspath(v)
from editor_amd import ops
out = {}
for dtype in (torch.float16, torch.bfloat16):
    b, t, heads, hd = 24, 129, 12, 64
    g = torch.Generator().manual_seed(1)
    qkv = (torch.randn(b * t, 3 * heads * hd, generator=g) * 0.8).to(dtype).cuda()
    do = (torch.randn(b * t, heads * hd, generator=g)).to(dtype).cuda()
    o, lse = ops.attention_fwd(qkv, b, t, heads, hd)
    dq = ops.attention_bwd(qkv, do, b, t, heads, hd, None, lse, o)
    lens = [129, 60, 1, 77, 128, 99, 140, 33]
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32).cuda()
    rows = (sum(lens) + 63) // 64 * 64
    q2 = qkv[:rows].clone(); q2[sum(lens):] = 0
    d2 = do[:rows].clone(); d2[sum(lens):] = 0
    o2, lse2 = ops.attention_fwd(q2, len(lens), max(lens), heads, hd, cu=cu)
    dq2 = ops.attention_bwd(q2, d2, len(lens), max(lens), heads, hd, None, lse2, o2, cu=cu)
    out[str(dtype)] = [x.cpu() for x in (o, lse, dq, o2, lse2, dq2)]
tag = "old" if v else "new"
torch.save(out, "gpurun_out/attn_%s.pt" % tag)
other = "gpurun_out/attn_%s.pt" % ("new" if v else "old")
if os.path.exists(other):
    ref = torch.load(other)
    for k in out:
        for name, a, b_ in zip(("o", "lse", "dqkv", "o_varlen", "lse_varlen", "dqkv_varlen"), out[k], ref[k]):
            same = torch.equal(a, b_)
            print(k, name, "bit-identical" if same else "DIFFERS max |d| %.3e (nan %d / %d)" % (
                (a.float() - b_.float()).abs().nan_to_num(0).max().item(), int(torch.isnan(a.float()).sum()), int(torch.isnan(b_.float()).sum())))
