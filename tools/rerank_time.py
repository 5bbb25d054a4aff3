"""Wall time of the device re-ranking (editor_amd.metrics.re_ranking) at the RGBNT100 test split's size (1 715 queries, 8 575 gallery
images, 2 304-wide features), k1 = 50, k2 = 15, lambda = 0.3 as utils/metrics.py:278 calls it."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editor_amd import metrics, synth  # noqa: E402

nq, ng, d = int(os.environ.get("NQ", 1715)), int(os.environ.get("NG", 8575)), 2304
ids = 50
pid = synth.integers(5, "rt/pid", (nq + ng,), ids)
proto = synth.normal(5, "rt/proto", (ids, d), 1.0)
feats = metrics.normalize((proto[pid] * 0.6 + synth.normal(5, "rt/noise", (nq + ng, d), 1.0)).cuda())
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = metrics.re_ranking(feats[:nq], feats[nq:], 50, 15, 0.3)
    torch.cuda.synchronize()
    print("re_ranking %d x %d: %.1f ms  (peak memory %.1f GB)" % (nq, ng, 1e3 * (time.perf_counter() - t0),
                                                                torch.cuda.max_memory_allocated() / 2 ** 30))
print(tuple(out.shape), float(out.min()), float(out.max()))
