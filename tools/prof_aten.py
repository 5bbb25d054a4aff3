"""Leftover torch / ATen launches of one training step in the bench configuration (gradient buckets on): per aten op (device time,
count, shapes) and - for every non-EDITOR kernel - the python-visible op it was launched under.   python tools/prof_aten.py"""
import contextlib
import io
import os
import sys
from collections import Counter

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editor_amd import config, losses, solver, synth  # noqa: E402
from editor_amd.modeling import make_model  # noqa: E402

cfg, num_class, cams = config.preset("RGBNT201", compute_dtype="bf16", drop_path=0.1)
torch.manual_seed(1111)
with contextlib.redirect_stdout(io.StringIO()):
    model = make_model(cfg, num_class, cams)
synth.fill_state_dict_(model.state_dict(), 1111)
model = model.cuda().train()
buckets = model.enable_grad_buckets()
opt, _ = solver.make_optimizer(cfg, model, None)
img, label, cam, view = synth.make_batch(1111, 128, 256, 128, cams, instances=16)
img = {k: v.cuda() for k, v in img.items()}
label, cam, view = label.cuda(), cam.cuda(), view.cuda()


class W:
    def add_scalar(self, *a, **k):
        pass


def step():
    opt.zero_grad(set_to_none=True)
    out = model(img, label=label, cam_label=cam, view_label=view, img_path=None, writer=W(), epoch=1)
    loss = losses.loss_pairs(out, label)
    loss.backward()
    buckets.finish()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:30]:
    print("%-28s n=%3d  dev %8.1f us  %s" % (e.key, e.count, e.device_time_total, str(e.input_shapes)[:110]))
# kernels that are not ours, attributed to the innermost CPU op that launched them
by = Counter()
tm = Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA or not ev.kernels:
        continue
    for k in ev.kernels:
        if "anonymous namespace" in k.name and "at::native" not in k.name:
            continue                                   # one of this repo's kernels
        short = k.name.split("<")[0][-50:] + ("<bf16>" if "BFloat16" in k.name else "")
        key = (short, ev.name, str(getattr(ev, "input_shapes", ""))[:70])
        by[key] += 1
        tm[key] += k.duration
print("\nnon-EDITOR kernels by launching op:")
for key, n in sorted(by.items(), key=lambda kv: -tm[kv[0]])[:40]:
    print("%4d x %8.1f us  %-52s <- %-24s %s" % (n, tm[key], key[0], key[1], key[2]))
