import sys, os, io, contextlib, torch
sys.path.insert(0, '.')
from editor_amd import config, losses, synth
from editor_amd.modeling import make_model
from editor_amd.optim import FusedSGD
cfg, num_class, cams = config.preset("RGBNT201", compute_dtype="bf16", drop_path=0.1)
torch.manual_seed(1111)
with contextlib.redirect_stdout(io.StringIO()):
    model = make_model(cfg, num_class, cams)
synth.fill_state_dict_(model.state_dict(), 1111)
model = model.cuda().train()
opt = FusedSGD(model.named_parameters(), base_lr=1e-3, weight_decay=1e-4, bias_lr_factor=2.0, weight_decay_bias=1e-4, momentum=0.9)
img, label, cam, view = synth.make_batch(1111, 128, 256, 128, cams, instances=16)
img = {k: v.cuda() for k, v in img.items()}; label, cam, view = label.cuda(), cam.cuda(), view.cuda()
class W:
    def add_scalar(self, *a, **k): pass
def step():
    opt.zero_grad(set_to_none=True)
    out = model(img, label=label, cam_label=cam, view_label=view, img_path=None, writer=W(), epoch=1)
    loss = losses.loss_pairs(out, label)
    loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:40]:
    print("%-28s n=%3d  dev %8.1f us  %s" % (e.key, e.count, e.device_time_total, str(e.input_shapes)[:110]))
