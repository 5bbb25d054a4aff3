import contextlib, io, sys, torch
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
    loss = losses.loss_pairs(out, label); loss.backward(); opt.step(); return loss
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): l0 = step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
opt.zero_grad(set_to_none=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): sl = step()
hist = [float(l0)]
for i in range(300):
    g.replay()
    if i % 50 == 49:
        torch.cuda.synchronize(); hist.append(float(sl))
print("loss trajectory (same synthetic batch, 300 graph-replayed SGD steps):", [round(h, 4) for h in hist])
bad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
print("non-finite parameters:", bad[:5], "peak mem GB %.1f" % (torch.cuda.max_memory_allocated() / 2**30))
