"""HIP-event timings of the selection kernels at the bench workload's sizes, over rotating operand sets (> 512 MB: HBM, not the
Infinity Cache):  python tools/select_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editor_amd import ops, synth  # noqa: E402


def ev(fn, nsets, reps=30):
    for i in range(nsets):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i % nsets)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for (h, w) in ((256, 128), (128, 256), (384, 128)):
    b = 128
    img, _, _, _ = synth.make_batch(1111, b, h, w, 4)
    sets = [[img[k].cuda().clone() for k in ("RGB", "NI", "TI")] for _ in range(5)]
    us = ev(lambda i: ops.freq_counts(*sets[i]), 5)
    nbytes = 3 * sets[0][0].numel() * 4
    print("freq_counts %dx%d B=%d: %.1f us  %.0f GB/s  frac %.3f" % (h, w, b, us, nbytes / us / 1e3, nbytes / us / 1e3 / 8000))
    one = sets[0]
    us1 = ev(lambda i: ops.freq_counts(*one), 1)
    print("   (one warm set: %.1f us  frac %.3f)" % (us1, nbytes / us1 / 1e3 / 8000))
counts = ops.freq_counts(*sets[0])
n = counts.shape[1]
us = ev(lambda i: ops.topk_mask(counts, 10), 1)
print("topk_mask<int> %d rows x %d, k=10: %.1f us" % (counts.shape[0], n, us))
scores = torch.rand(3 * 128 * 12, 128, device="cuda") * 0.01
us = ev(lambda i: ops.topk_mask(scores, 2, group=12), 1)
print("topk_mask<float> %d rows x 128, k=2: %.1f us" % (scores.shape[0], us))
scores = torch.rand(4 * 64 * 16, 512, device="cuda") * 0.01
us = ev(lambda i: ops.topk_mask(scores, 2, group=16), 1)
print("topk_mask<float> %d rows x 512, k=2: %.1f us" % (scores.shape[0], us))
