"""What rounding every residual branch (projection and fc2 outputs) to 16 bits BEFORE the residual add would cost in accuracy - the
accuracy side of the "bf16-branch residual" priced in tools/residual_pricing.py.  The fused kernels are emulated
with torch ops on top of the product path (EDITOR_BRANCH16_EMULATE: the GEMM writes the branch in the activation dtype with its plain
epilogue, the add happens in fp32 afterwards); eval forward at B = 128, config 2, selection teacher-forced, against the oracle.
    python tools/branch16_accuracy.py [bf16|f16]"""
import contextlib
import io
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editor_amd import config, functional as fn, ops, synth  # noqa: E402
from editor_amd.modeling import make_model  # noqa: E402
from oracle import editor_ref as oracle  # noqa: E402   (a tool: the oracle is the checker here)

dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = 128
torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
cfg, c, cams = config.preset("RGBNT201", compute_dtype=dt, drop_path=0.0)
with contextlib.redirect_stdout(io.StringIO()):
    m = make_model(cfg, c, cams)
synth.fill_state_dict_(m.state_dict(), 61)
sd = {k: v.clone() for k, v in m.state_dict().items()}
img, label, cam, view = synth.make_batch(62, B, 256, 128, cams, instances=16)
with torch.no_grad():
    ref, aux = oracle.editor_forward(sd, img, cam, training=False, al=cfg.MODEL.AL, return_aux=True)
m = m.cuda().eval()
m.teacher_index = aux["index"]
gimg = {k: v.cuda() for k, v in img.items()}


def run():
    with torch.no_grad():
        out = m(gimg, cam_label=cam.cuda(), view_label=view.cuda())
    return ((out.cpu().double() - ref.double()).norm() / ref.double().norm()).item()


base = run()
# emulate: residual epilogue -> plain 16-bit output + fp32 add afterwards
orig = ops.gemm


def gemm16(a, b, c_, m_, n, k, *args, **kw):
    if kw.get("epilogue", 0) == ops.EPI_RESIDUAL and a.dtype in ops.HALF_DTYPES and c_.dtype == torch.float32:
        br = torch.empty(c_.shape, dtype=a.dtype, device=a.device)
        kw2 = dict(kw)
        aux_, rs = kw2.pop("aux"), kw2.pop("rowscale", None)
        kw2["epilogue"] = 0
        orig(a, b, br, m_, n, k, *args, **kw2)
        x = br.float()
        if rs is not None:
            x = x * rs[:, None]
        c_.copy_(aux_ + x)
        return
    return orig(a, b, c_, m_, n, k, *args, **kw)


ops.gemm = gemm16
emu = run()
print("%s eval B=%d cls4t rel err: product path %.4g   with every branch rounded to %s before the add %.4g   (x %.3f)" % (dt, B, base, dt, emu, emu / base))
