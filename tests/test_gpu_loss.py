"""Row N1 (loss head) parity: HIP CrossEntropyLabelSmooth + batch-hard soft-margin triplet through the C ABI against
(i) the golden captured from the reference's layers/make_loss.py (tests/golden/f7_loss.npz) and (ii) the oracle on
seeded inputs, including strided feature views, a single-instance identity and the engine's loss_pairs assembly."""
import pytest
import torch

from conftest import load_golden, rel_err
from editor_amd import synth

pytestmark = pytest.mark.gpu


def test_loss_head_matches_reference_golden():
    from editor_amd import losses
    g = load_golden("f7_loss")
    seed = int(g["seed"])
    score = synth.normal(seed, "loss/score", (32, 171), 2.0).cuda().requires_grad_(True)
    feat = synth.normal(seed, "loss/feat", (32, 2304), 1.0).cuda().requires_grad_(True)
    target = torch.arange(4).repeat_interleave(8).cuda()
    loss_fn, center = losses.make_loss(None, num_classes=171)
    assert center.centers.shape == (171, 2048)
    loss = loss_fn(score=score, feat=feat, target=target, target_cam=None)
    loss.backward()
    assert rel_err(loss.detach().cpu(), g["loss"]) < 1e-6
    assert rel_err(score.grad.cpu(), g["dscore"]) < 1e-5
    assert rel_err(feat.grad[:, :64].cpu(), g["dfeat"]) < 1e-5
    assert abs(feat.grad.norm().item() / float(g["dfeat_norm"]) - 1) < 1e-5


@pytest.mark.parametrize("b,k,c,d", [(128, 8, 201, 2304), (64, 4, 100, 768), (6, 2, 7, 32), (256, 16, 1501, 2304)])
def test_loss_terms_match_oracle(oracle, b, k, c, d):
    from editor_amd import losses
    seed = 60 + b
    score = synth.normal(seed, "s", (b, c), 3.0)
    wide = synth.normal(seed, "f", (b, d + 16), 0.7)            # features arrive as a strided column slice
    target = torch.arange(b // k).repeat_interleave(k)
    target = target[synth.integers(seed, "perm", (b,), 1 << 30).argsort()]
    up = 0.37
    sr, wr = score.clone().requires_grad_(True), wide.clone().requires_grad_(True)
    (oracle.cross_entropy_label_smooth(sr, target) * up).backward()
    lt_ref = oracle.triplet_soft_margin(wr[:, 8:8 + d], target)
    (lt_ref * up).backward()
    sg, wg = score.cuda().requires_grad_(True), wide.cuda().requires_grad_(True)
    lc = losses.cross_entropy_label_smooth(sg, target.cuda())
    lt = losses.triplet_soft_margin(wg[:, 8:8 + d], target.cuda())
    ((lc + lt) * up).backward()
    assert rel_err(lc.detach().cpu(), oracle.cross_entropy_label_smooth(score, target)) < 1e-6
    assert rel_err(lt.detach().cpu(), lt_ref.detach()) < 1e-5
    assert rel_err(sg.grad.cpu(), sr.grad) < 1e-5
    assert rel_err(wg.grad.cpu(), wr.grad) < 2e-5


def test_loss_pairs_assembly_and_determinism(oracle):
    from editor_amd import losses
    b = 32
    target = torch.arange(4).repeat_interleave(8)
    outs = []
    for i in range(4):
        outs += [synth.normal(7, f"s{i}", (b, 50), 2.0), synth.normal(7, f"f{i}", (b, 768 * (1 + 2 * (i == 0))), 1.0)]
    aux = torch.tensor(0.25)
    ref = oracle.loss_pairs(tuple(outs) + (aux,), target)
    dev = [o.cuda().requires_grad_(True) for o in outs]
    got = losses.loss_pairs(tuple(dev) + (aux.cuda(),), target.cuda())
    got.backward()
    assert rel_err(got.detach().cpu(), ref) < 1e-6
    g1 = [d.grad.clone() for d in dev]
    for d in dev:
        d.grad = None
    losses.loss_pairs(tuple(dev) + (aux.cuda(),), target.cuda()).backward()
    assert all(torch.equal(a, d.grad) for a, d in zip(g1, dev))      # fixed-order reductions: bit-reproducible


def test_loss_rejects_cpu_tensors():
    from editor_amd import losses
    with pytest.raises(Exception):
        losses.cross_entropy_label_smooth(torch.randn(4, 5), torch.tensor([0, 1, 2, 3]))


def test_center_loss_vs_reference_golden_and_oracle(oracle):
    """CenterLoss (layers/center_loss.py:30-51) on the device: the reference's own value / gradients (golden f18), then a second shape
    against the oracle, through the module with the reference's name (editor_amd.losses.CenterLoss)."""
    from conftest import load_golden
    from editor_amd import losses
    g = load_golden("f18_center_loss")
    seed = int(g["seed"])
    b, c, d = (int(v) for v in g["shape"])
    x = synth.normal(seed, "cl/x", (b, d), 1.0).cuda().requires_grad_(True)
    lab = torch.from_numpy(g["label"]).cuda()
    cl = losses.CenterLoss(num_classes=c, feat_dim=d).cuda()
    with torch.no_grad():
        cl.centers.copy_(synth.normal(seed, "cl/c", (c, d), 1.0))
    loss = cl(x, lab)
    assert rel_err(loss.detach().cpu(), g["loss"]) < 1e-5
    (3.0 * loss).backward()
    assert rel_err(x.grad[:, :64].cpu(), g["dx"]) < 1e-5 and abs(x.grad.norm().item() / float(g["dx_norm"]) - 1) < 1e-5
    uniq = lab.unique()
    assert rel_err(cl.centers.grad[uniq][:, :64].cpu(), g["dc"]) < 1e-5
    assert abs(cl.centers.grad.norm().item() / float(g["dc_norm"]) - 1) < 1e-5
    # B = 128, 2304-wide features (cls4t), 171 classes
    x2 = synth.normal(7, "cl2/x", (128, 2304), 1.0)
    c2 = synth.normal(7, "cl2/c", (171, 2304), 1.0)
    lab2 = torch.arange(8).repeat_interleave(16) * 5
    xr, cr = x2.clone().requires_grad_(True), c2.clone().requires_grad_(True)
    lr = oracle.center_loss(xr, cr, lab2)
    lr.backward()
    cl2 = losses.CenterLoss(num_classes=171, feat_dim=2304).cuda()
    with torch.no_grad():
        cl2.centers.copy_(c2)
    xg = x2.cuda().requires_grad_(True)
    lg = cl2(xg, lab2.cuda())
    lg.backward()
    assert rel_err(lg.detach().cpu(), lr.detach()) < 1e-5
    assert rel_err(xg.grad.cpu(), xr.grad) < 1e-5 and rel_err(cl2.centers.grad.cpu(), cr.grad) < 1e-5
    with pytest.raises(RuntimeError):
        losses.CenterLoss(4, 8)(torch.zeros(2, 8), torch.zeros(2, dtype=torch.long))          # CPU tensors: no fallback
