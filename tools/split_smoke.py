import torch, sys, time
sys.path.insert(0, "/root/repo")
from editor_amd import config, synth
from editor_amd.modeling import make_model
from oracle import editor_ref as oracle
dev = torch.device("cuda", 0)
seed, b = 7, 2
cfg, c, cams = config.preset("RGBNT201", compute_dtype="f16x2", drop_path=0.0)
m = make_model(cfg, c, cams)
synth.fill_state_dict_(m.state_dict(), seed)
sd = {k: v.clone() for k, v in m.state_dict().items()}
img, label, cam, view = synth.make_batch(seed, b, 256, 128, cams, instances=1)
with torch.no_grad():
    ref, aux = oracle.editor_forward(sd, img, cam, training=False, al=cfg.MODEL.AL, return_aux=True)
m = m.to(dev).eval()
gimg = {k: v.to(dev) for k, v in img.items()}
with torch.no_grad():
    out = m(gimg, cam_label=cam.to(dev), view_label=view.to(dev))
print("index equal:", torch.equal(m.last_aux["index"].cpu().bool(), aux["index"]))
sc = m.last_aux["scores"].cpu(); osc = torch.stack(list(aux["scores"])) if isinstance(aux["scores"], (list, tuple)) else aux["scores"]
print("scores rel err", ((sc.view(-1) - osc.reshape(-1)).norm() / osc.norm()).item())
print("cls4t rel err", ((out.cpu() - ref).norm() / ref.norm()).item())
for dt in ("f32", "f16"):
    cfg2, c, cams = config.preset("RGBNT201", compute_dtype=dt, drop_path=0.0)
    m2 = make_model(cfg2, c, cams); m2.load_state_dict(sd); m2 = m2.to(dev).eval()
    m2.teacher_index = aux["index"]
    with torch.no_grad():
        o2 = m2(gimg, cam_label=cam.to(dev), view_label=view.to(dev))
    print(dt, "cls4t rel err", ((o2.cpu() - ref).norm() / ref.norm()).item())
