"""Upper-bound probe: is it faster to push two independent half-batches (B/2 each) through the model on two streams
than one batch of B?  (Semantics differ - BN / triplet per half - this only measures GPU throughput.)"""
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
class W:
    def add_scalar(self, *a, **k): pass
def batch(b, seed):
    img, label, cam, view = synth.make_batch(seed, b, 256, 128, cams, instances=16)
    return {k: v.cuda() for k, v in img.items()}, label.cuda(), cam.cuda(), view.cuda()
full = batch(128, 1)
halves = [batch(64, 2), batch(64, 3)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def step_full():
    opt.zero_grad(set_to_none=True)
    img, label, cam, view = full
    out = model(img, label=label, cam_label=cam, view_label=view, img_path=None, writer=W(), epoch=1)
    losses.loss_pairs(out, label).backward(); opt.step()
def step_two():
    opt.zero_grad(set_to_none=True)
    ls = []
    cur = torch.cuda.current_stream()
    for st, (img, label, cam, view) in zip(streams, halves):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            out = model(img, label=label, cam_label=cam, view_label=view, img_path=None, writer=W(), epoch=1)
            ls.append(losses.loss_pairs(out, label))
    for st in streams: cur.wait_stream(st)
    (ls[0] + ls[1]).backward()
    for st in streams: cur.wait_stream(st)
    opt.step()
def bench(fn, n=6):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("one batch of 128        : %.2f ms/step" % bench(step_full))
print("two half-batches, 2 strm: %.2f ms/step" % bench(step_two))
