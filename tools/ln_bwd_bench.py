"""us per launch of the LayerNorm backward (plain and cast form) at the step's size, operands rotating over 4 sets (> 512 MB each way)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editor_amd import ops  # noqa: E402

m, d = 3 * 128 * 129, 768
g = torch.Generator(device="cuda").manual_seed(0)
sets = []
for _ in range(4):
    x = torch.randn(m, d, device="cuda", generator=g)
    dy = torch.randn(m, d, device="cuda", generator=g).bfloat16()
    dxin = torch.randn(m, d, device="cuda", generator=g)
    mean, rstd = x.mean(1), 1.0 / x.std(1)
    sets.append((x, dy, dxin, mean.contiguous(), rstd.contiguous(), torch.rand(m, device="cuda", generator=g)))
gamma = torch.randn(d, device="cuda", generator=g)


def bench(fn, reps=40):
    for i in range(4):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i % 4)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


plain = lambda i: ops.layernorm_bwd(sets[i][1], sets[i][0], gamma, sets[i][3], sets[i][4], dx_in=sets[i][2])
cast = lambda i: ops.layernorm_bwd_cast(sets[i][1], sets[i][0], gamma, sets[i][3], sets[i][4], sets[i][2], sets[i][5])
nores = lambda i: ops.layernorm_bwd(sets[i][1], sets[i][0], gamma, sets[i][3], sets[i][4])
for name, fn, bytes_ in (("plain + residual gradient", plain, m * d * (2 + 4 + 4 + 4)), ("cast form", cast, m * d * (2 + 4 + 4 + 4 + 2)),
                         ("no residual gradient", nores, m * d * (2 + 4 + 4))):
    t = min(bench(fn) for _ in range(3))
    print("%-28s %7.1f us  %5.2f TB/s" % (name, t, bytes_ / t / 1e6))
