"""How long does the HOST need to issue one training step (no device sync inside)?  If this approaches the GPU time of
the step, the bench becomes launch-bound on a busy host."""
import contextlib, io, sys, time, torch
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
tm = {"fwd": 0.0, "loss": 0.0, "bwd": 0.0, "opt": 0.0}
def step():
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    out = model(img, label=label, cam_label=cam, view_label=view, img_path=None, writer=W(), epoch=1)
    t1 = time.perf_counter()
    loss = losses.loss_pairs(out, label)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    tm["fwd"] += t1 - t0; tm["loss"] += t2 - t1; tm["bwd"] += t3 - t2; tm["opt"] += t4 - t3
for _ in range(3): step()
torch.cuda.synchronize()
ts = []
for k in tm: tm[k] = 0.0
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print("host issue ms / step wall ms:", [(round(a * 1e3, 1), round(b * 1e3, 1)) for a, b in ts])
print("host ms per step by phase:", {k: round(v / 5 * 1e3, 2) for k, v in tm.items()})
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); step(); pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)
