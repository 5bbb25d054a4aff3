"""Runs the HBM-bound selection kernels once at the bench workload's sizes (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editor_amd import ops, synth  # noqa: E402

b = 128
img, _, _, _ = synth.make_batch(1111, b, 256, 128, 4)
rgb, nir, tir = (img[k].cuda() for k in ("RGB", "NI", "TI"))
probs = torch.rand(12, 3 * b, 12, 129, 132, device="cuda")
feat = torch.randn(3, b, 129, 768, device="cuda")
for _ in range(3):
    mask, counts = ops.frequency_mask(rgb, nir, tir, 10)
    scores = ops.attn_rollout(probs)
    m = ops.topk_mask(scores.view(-1, 128), 2, group=12)
    out, loss = ops.sfts_apply(feat, mask, True)
torch.cuda.synchronize()
print("ok", int(counts.sum()), float(loss))
