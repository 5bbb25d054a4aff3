"""Wall-clock (HIP events, eager) of the forward stages of one training step: patch embedding + backbone, token selection,
SFTS apply, HMA head, heads + loss; and of the whole backward + optimizer.   python tools/stage_times.py [bf16|f16]"""
import contextlib, io, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editor_amd import config, losses, synth
from editor_amd.modeling import make_model
from editor_amd.optim import FusedSGD

dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
cfg, num_class, cams = config.preset("RGBNT201", compute_dtype=dt, drop_path=0.1)
torch.manual_seed(1111)
with contextlib.redirect_stdout(io.StringIO()):
    model = make_model(cfg, num_class, cams)
synth.fill_state_dict_(model.state_dict(), 1111)
model = model.cuda().train()
opt = FusedSGD(model.named_parameters(), base_lr=1e-3, weight_decay=1e-4, bias_lr_factor=2.0, weight_decay_bias=1e-4, momentum=0.9)
img, label, cam, view = synth.make_batch(1111, 128, 256, 128, cams, instances=16)
img = {k: v.cuda() for k, v in img.items()}; label, cam, view = label.cuda(), cam.cuda(), view.cuda()
marks = []
def wrap(name):
    f = getattr(model, name)
    def g(*a, **k):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); r = f(*a, **k); e1.record(); marks.append((name, e0, e1)); return r
    setattr(model, name, g)
for n in ("_backbone", "_select", "_hma_compact"):
    wrap(n)
class W:
    def add_scalar(self, *a, **k): pass
def step(timed=False):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    opt.zero_grad(set_to_none=True)
    ev[0].record()
    out = model(img, label=label, cam_label=cam, view_label=view, img_path=None, writer=W(), epoch=1)
    loss = losses.loss_pairs(out, label)
    ev[1].record()
    loss.backward()
    ev[2].record()
    opt.step()
    ev[3].record()
    return ev
for _ in range(3): step()
res = {}
for it in range(5):
    marks.clear()
    ev = step(); torch.cuda.synchronize()
    for n, a, b in marks: res.setdefault(n, []).append(a.elapsed_time(b))
    res.setdefault("forward+loss", []).append(ev[0].elapsed_time(ev[1]))
    res.setdefault("backward", []).append(ev[1].elapsed_time(ev[2]))
    res.setdefault("optimizer", []).append(ev[2].elapsed_time(ev[3]))
for k, v in res.items():
    print("%-14s %7.2f ms" % (k, sorted(v)[len(v) // 2]))
