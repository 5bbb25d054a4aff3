"""The whole per-batch path a training loop built on this repo runs, end to end on the device, from FILE BYTES to the
optimizer update - the pieces of SURVEY.md 8 rows N3 -> A1..A9 -> N1 -> N4 wired the way INTEGRATION.md shows:

    JPEG bytes --DeviceJpegDecoder--> uint8 crops (RGB | NI | TI) --DeviceResize--> --DeviceTrainTransform--> fp32 (B,3,H,W)
      --make_model(...)(img, label, cam_label, ...)--> (score, feat) pairs + aux loss --make_loss / loss_pairs--> loss
      --DeviceGradScaler.scale(loss).backward()--> gradient slots (GradBuckets, no process group) --scaler.step(FusedSGD)-->
      --scaler.update()

in the f16 mode (the reference's autocast dtype, engine/processor.py:79,94-96), eager and as a hipGraph replay of the model
step.  Checks: the decode feeding it is Pillow's pixels (golden), losses finite and decreasing on a repeated batch, an injected
overflow leaves the weights untouched and halves the device-resident scale, the captured step equals the eager one."""
import os

import numpy as np
import pytest
import torch

from editor_amd import config, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


class _Writer:
    def add_scalar(self, *a, **k):
        pass


def _pipeline(dtype="f16", batch=12):
    import contextlib
    import io
    from editor_amd import losses, solver
    from editor_amd.data import DeviceJpegDecoder, DeviceResize, DeviceTrainTransform
    from editor_amd.modeling import make_model
    from editor_amd.optim import DeviceGradScaler
    import random
    torch.manual_seed(3)
    random.seed(3)                                                 # (the erasing rectangles come from `random`, like the reference's)
    cfg, c, cams = config.preset("RGBNT201", compute_dtype=dtype, drop_path=0.1, grad_scale=1.0)
    with contextlib.redirect_stdout(io.StringIO()):
        model = make_model(cfg, c, cams)
    synth.fill_state_dict_(model.state_dict(), 17)
    model = model.cuda().train()
    buckets = model.enable_grad_buckets()                         # in-place gradient slots; no process group -> no collective
    opt, _ = solver.make_optimizer(cfg, model, None)
    scaler = DeviceGradScaler("cuda", init_scale=2.0 ** 12, growth_interval=1000)
    loss_fn, _ = losses.make_loss(cfg, c)
    g = np.load(os.path.join(HERE, "golden", "f14_decode.npz"))
    three = [g[n + ".jpg"].tobytes() for n in ("stitched_420_q75", "stitched_444_q90", "stitched_422_q85")]
    files = (three * ((batch + 2) // 3))[:batch]
    dec = DeviceJpegDecoder(crop_w=256, threads=4)
    rs = DeviceResize(cfg.INPUT.SIZE_TRAIN, interpolation=3)
    tf = DeviceTrainTransform(cfg.INPUT.SIZE_TRAIN)
    crops = dec(files, "cuda")                                     # (3, 12, 128, 256, 3) uint8
    assert np.array_equal(crops[1, 0].cpu().numpy(), g["stitched_420_q75.rgb"][:, 256:512])       # Pillow's pixels
    params = tf.draw(len(files))
    img = {k: tf(rs(crops[i]), params=params, seed=5 + i) for i, k in enumerate(("RGB", "NI", "TI"))}
    assert img["RGB"].shape == (batch, 3, 256, 128) and img["RGB"].dtype == torch.float32
    label = torch.arange(batch // 4).repeat_interleave(4).cuda()   # identities x 4 instances
    cam = torch.zeros(batch, dtype=torch.int64).cuda()

    def step(poison=None):
        opt.zero_grad()
        out = model(img, label=label, cam_label=cam, view_label=cam, writer=_Writer(), epoch=1)
        loss = losses.loss_pairs(out, label, loss_fn)
        if poison is not None:
            loss = loss * poison
        scaler.scale(loss).backward()
        buckets.finish()
        scaler.step(opt)
        scaler.update()
        return loss.detach()
    return model, opt, scaler, step


def test_bytes_to_update_f16_with_device_grad_scaler():
    model, opt, scaler, step = _pipeline("f16")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        l0 = float(step())
        w_before = model.BACKBONE.base.blocks[3].mlp.fc1.weight.detach().clone()
        l1 = float(step())
        l2 = float(step())
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert all(np.isfinite(v) for v in (l0, l1, l2)) and l2 < l0              # the repeated batch is being fitted
    assert not torch.equal(model.BACKBONE.base.blocks[3].mlp.fc1.weight.detach(), w_before)
    assert scaler.get_scale() == 2.0 ** 12 and not opt.found_inf()
    # an overflow in the backward: nothing is written, the scale backs off
    # (trainable parameters: the OCFR centre tables are requires_grad=False Parameters that the FORWARD updates, OCFR.py:80-83)
    keep = {k: v.detach().clone() for k, v in model.named_parameters() if v.requires_grad}
    with torch.cuda.stream(side):
        step(poison=torch.tensor(float("inf"), device="cuda"))
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    changed = [k for k, v in model.named_parameters() if v.requires_grad and not torch.equal(v.detach(), keep[k])]
    assert not changed, changed[:6]
    assert scaler.get_scale() == 2.0 ** 11 and opt.found_inf()


def test_captured_step_with_device_grad_scaler_equals_eager():
    """Two models from the same seed: three eager steps vs one eager + a captured step replayed twice - identical parameters
    and identical device-resident scale (everything GradScaler does happens on the device).  B = 64: 3 * 64 * 129 token rows
    are a multiple of 64, so every weight gradient takes the deterministic slab path (a ragged tail goes through fp32 atomics)."""
    m1, o1, s1, step1 = _pipeline("f16", 64)
    m2, o2, s2, step2 = _pipeline("f16", 64)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step1()
        step2()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    # bring model 2 to the same drop-path / BN state, then capture its step
    o2.zero_grad()
    torch.cuda.empty_cache()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step2()
    graph.replay()
    graph.replay()
    torch.cuda.synchronize()
    sd1, sd2 = m1.state_dict(), m2.state_dict()
    diff = [k for k in sd1 if not torch.equal(sd1[k], sd2[k])]
    assert not diff, diff[:6]
    assert s1.get_scale() == s2.get_scale()
