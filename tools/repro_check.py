import sys, os, io, contextlib, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from editor_amd import config, synth, losses
from editor_amd.optim import FusedSGD
from editor_amd.modeling import make_model
class Q:
    def add_scalar(self, *a, **k): pass
def build():
    torch.manual_seed(77)
    cfg, c, cams = config.preset("RGBNT201", compute_dtype="bf16", drop_path=0.1)
    with contextlib.redirect_stdout(io.StringIO()):
        m = make_model(cfg, c, cams)
    synth.fill_state_dict_(m.state_dict(), 31)
    m = m.cuda().train()
    opt = FusedSGD(m.named_parameters(), base_lr=1e-2, weight_decay=1e-4, bias_lr_factor=2.0, weight_decay_bias=1e-4, momentum=0.9)
    return m, opt, cams
b = int(os.environ.get("DBG_B", "64"))
m, o, cams = build()
img, label, cam, view = synth.make_batch(5, b, 256, 128, cams, instances=8)
img = {k: v.cuda() for k, v in img.items()}; label, cam, view = label.cuda(), cam.cuda(), view.cuda()
def mk(m, opt):
    def step():
        opt.zero_grad(set_to_none=True)
        out = m(img, label=label, cam_label=cam, view_label=view, img_path=None, writer=Q(), epoch=1)
        loss = losses.loss_pairs(out, label); loss.backward(); opt.step(); return loss
    return step
side = torch.cuda.Stream()
def run_eager(n):
    m, o, _ = build(); s = mk(m, o)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(n): l = s()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    return m, float(l)
def run_graph(w, r):
    m, o, _ = build(); s = mk(m, o)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(w): s()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    o.zero_grad(set_to_none=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): l = s()
    for _ in range(r): g.replay()
    torch.cuda.synchronize()
    return m, float(l)
keys = ["BACKBONE.base.blocks.3.attn.qkv.weight", "BACKBONE.base.blocks.11.mlp.fc2.bias", "FUSE_HEAD.weight", "FUSE_block.attn1.qkv.weight", "BACKBONE.base.cls_token"]
def cmp(a, b, tag):
    sa, sb = a.state_dict(), b.state_dict()
    print(tag, {k[-22:]: float((sa[k] - sb[k]).abs().max()) for k in keys})
for n in (1, 2):
    e1, l1 = run_eager(n); e2, l2 = run_eager(n)
    cmp(e1, e2, f"eager{n} vs eager{n} (loss {l1:.5f} {l2:.5f})")
os.environ["X"] = "1"
e5, l5 = run_eager(5); g5, lg = run_graph(2, 3)
cmp(e5, g5, f"eager5 vs 2+3 graph (loss {l5:.5f} {lg:.5f})")
g5b, _ = run_graph(2, 3)
cmp(g5, g5b, "graph vs graph")
