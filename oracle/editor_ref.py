"""ORACLE - test infrastructure only.

CPU (plain PyTorch fp32 + a small C library for top-k tie order) restatement of the
reference's per-batch forward for the EDITOR hot path.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this; the product path (editor_amd/) never does.

Everything is a pure function over a name-compatible state dict `sd`, so the
same seeded weights drive the reference module, this oracle and the HIP model.  Backward is
torch autograd over these functions (the selection stages are non-differentiable, as in the
reference).  Each function cites the reference lines it restates.

Parity pinning: tests/golden/*.npz were produced by the REFERENCE itself, imported in the build
container with the shims of tools/ref_shims.py (tests/golden/capture_golden.py);
tests/test_oracle_golden.py checks this file against them (masks bit-exact, floats <= 1e-5 rel).
"""
import ctypes
import math
import os

import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libeditor_oracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _LIB = ctypes.CDLL(path)
    return _LIB


# ----------------------------------------------------------------------------------------------
# top-k with torch.topk's CPU tie order (SFTS.py:155, Frequency.py:58; oracle/topk_ref.c)
# ----------------------------------------------------------------------------------------------
def topk_indices(x, k):
    """x: (R, n) float32 or int32 CPU tensor -> (R, k) int64 in torch.topk order."""
    x = x.detach().contiguous().cpu()
    rows, n = x.shape
    out = torch.empty(rows, k, dtype=torch.int64)
    if x.dtype == torch.int32:
        fn = _lib().editor_oracle_topk_i32
    else:
        x = x.float().contiguous()
        fn = _lib().editor_oracle_topk_f32
    rc = fn(ctypes.c_void_p(x.data_ptr()), ctypes.c_long(rows), ctypes.c_long(n),
            ctypes.c_long(k), ctypes.c_void_p(out.data_ptr()))
    if rc != 0:
        raise ValueError("topk: bad arguments")
    return out


def topk_mask(x, k):
    """Bool (R, n) mask of the k selected positions: topk -> sort -> scatter
    (SFTS.py:155-158, Frequency.py:58-62)."""
    idx = topk_indices(x, k)
    mask = torch.zeros(x.shape, dtype=torch.bool)
    mask.scatter_(1, idx, True)
    return mask


# ----------------------------------------------------------------------------------------------
# A2  frequency branch (Frequency.py:65-84,42-63; pytorch_wavelets/dwt/lowlevel.py:91-172,226-271)
# ----------------------------------------------------------------------------------------------
_S = torch.tensor(1.0 / math.sqrt(2.0), dtype=torch.float32)   # pywt haar taps, cast to fp32


def _analysis_1d(x, dim):
    """afb1d, haar, mode='zero', even length (filters reversed by prep_filt_afb1d, lowlevel.py:970-974;
    grouped conv2d stride 2, no padding, lowlevel.py:164).  torch's CPU conv2d (oneDNN) evaluates the
    2-tap dot product as round(s*x[2i]) followed by ONE fused multiply-add with x[2i+1]:
        lo = fma(s, x[2i+1], s*x[2i])      hi = fma(-s, x[2i+1], s*x[2i])
    (measured against the reference in the build container, bit-exact on 3072/3072 outputs, both dims;
    the un-fused s*a + s*b differs in ~1/3 of the outputs by 1 ulp).  fma emulated in float64."""
    even = x.index_select(dim, torch.arange(0, x.shape[dim], 2))
    odd = x.index_select(dim, torch.arange(1, x.shape[dim], 2))
    p = (_S * even).double()
    q = _S.double() * odd.double()
    return (p + q).float(), (p - q).float()


def _synthesis_1d(lo, hi, dim):
    """sfb1d, haar: y[2i] = s*lo + s*hi ; y[2i+1] = s*lo - s*hi (conv_transpose2d stride 2,
    lowlevel.py:262-267)."""
    a = _S * lo + _S * hi
    b = _S * lo - _S * hi
    y = torch.stack([a, b], dim=dim + 1 if dim >= 0 else dim)
    shape = list(lo.shape)
    shape[dim] *= 2
    return y.reshape(shape)


def haar_dwt2(x, levels=4):
    """DWTForward(J, 'haar', 'zero') (transform2d.py:44-74; AFB2D lowlevel.py:336-347):
    rows (dim 3) then columns (dim 2); bands ordered [row-lo/col-hi, row-hi/col-lo, hi/hi]."""
    highs = []
    ll = x
    for _ in range(levels):
        rlo, rhi = _analysis_1d(ll, 3)
        ll, lh = _analysis_1d(rlo, 2)
        hl, hh = _analysis_1d(rhi, 2)
        highs.append(torch.stack([lh, hl, hh], dim=2))
    return ll, highs


def haar_idwt2(ll, highs):
    """DWTInverse('haar','zero') (transform2d.py:111-148; SFB2D lowlevel.py:671-680)."""
    for h in highs[::-1]:
        lh, hl, hh = h.unbind(dim=2)
        lo = _synthesis_1d(ll, lh, 2)
        hi = _synthesis_1d(hl, hh, 2)
        ll = _synthesis_1d(lo, hi, 3)
    return ll


def frequency_counts(rgb, nir, tir, window=16, levels=4, extra=()):
    """Per-patch count of positive pixels of IDWT(mean_m DWT(x_m)) averaged over channels
    (Frequency.py:65-80 and :42-56).  Returns int32 (B, H/window * W/window), row-major patches.
    extra: further modalities (the 4-modal extension of BASELINE config 5 - the same mean over one more term)."""
    mods = [m for m in (rgb, nir, tir) + tuple(extra) if m is not None]
    coeffs = [haar_dwt2(m.float(), levels) for m in mods]
    nm = float(len(mods))
    low = sum(c[0] for c in coeffs) / nm
    high = [sum(c[1][j] for c in coeffs) / nm for j in range(levels)]
    inv = haar_idwt2(low, high).mean(dim=1)                       # (B,H,W)
    b, h, w = inv.shape
    pos = inv.gt(0).reshape(b, h // window, window, w // window, window)
    return pos.sum(dim=(2, 4)).to(torch.int32).reshape(b, -1), inv


def frequency_mask(rgb, nir, tir, keep=10, window=16, extra=()):
    counts, _ = frequency_counts(rgb, nir, tir, window, extra=extra)
    return topk_mask(counts, int(keep)), counts


# ----------------------------------------------------------------------------------------------
# A3  ViT backbone (vit_pytorch.py:623-644, 449-458, 184-198, 139-145, 215-220)
# ----------------------------------------------------------------------------------------------
def _ln(x, sd, prefix, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def _drop_path(x, keep_mask, keep_prob):
    """drop_path with a teacher-forced per-sample keep mask (vit_pytorch.py:52-69): `x.div(keep_prob) * random_tensor`,
    random_tensor = floor(keep_prob + U[0,1)) of shape (B,1,1) - here handed in as the 0/1 mask the draw produced."""
    if keep_mask is None:
        return x
    return x.div(keep_prob) * keep_mask.view(-1, 1, 1)


def vit_block(x, sd, p, heads, eps=1e-6, keep=None, keep_prob=1.0, qk_scale=None):
    """Block.forward(get_att=True) (vit_pytorch.py:215-220) with Attention (:184-198), Mlp (:139-145).
    keep: None or a (2, B) pair of 0/1 masks - the reference calls self.drop_path TWICE per block (:217 attention branch,
    :218 MLP branch), each call drawing its own torch.rand((B,1,1))."""
    b, t, d = x.shape
    hd = d // heads
    keep_a, keep_m = (None, None) if keep is None else (keep[0], keep[1])
    h = _ln(x, sd, p + ".norm1", eps)
    qkv = F.linear(h, sd[p + ".attn.qkv.weight"], sd.get(p + ".attn.qkv.bias"))
    qkv = qkv.reshape(b, t, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * (qk_scale or hd ** -0.5)       # vit_pytorch.py:176: qk_scale or head_dim ** -0.5
    attn = attn.softmax(dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(b, t, d)
    o = F.linear(o, sd[p + ".attn.proj.weight"], sd.get(p + ".attn.proj.bias"))
    x = x + _drop_path(o, keep_a, keep_prob)
    h = _ln(x, sd, p + ".norm2", eps)
    h = F.linear(h, sd[p + ".mlp.fc1.weight"], sd.get(p + ".mlp.fc1.bias"))
    h = F.gelu(h)                                            # nn.GELU() default = exact erf
    h = F.linear(h, sd[p + ".mlp.fc2.weight"], sd.get(p + ".mlp.fc2.bias"))
    x = x + _drop_path(h, keep_m, keep_prob)
    return x, attn


def vit_forward(sd, img, cam, heads=12, sie_coef=3.0, prefix="BACKBONE.base", drop_keep=None,
                drop_rates=None, qk_scale=None):
    """Trans.forward (vit_pytorch.py:623-644).  drop_keep: optional (depth, 2, B) 0/1 keep masks - [layer, branch
    (0 = attention, 1 = MLP), sample]; a block whose rate is 0 holds nn.Identity (:209) and draws nothing."""
    w = sd[prefix + ".patch_embed.proj.weight"]
    x = F.conv2d(img, w, sd[prefix + ".patch_embed.proj.bias"], stride=w.shape[-1])
    x = x.flatten(2).transpose(1, 2)
    b = x.shape[0]
    x = torch.cat([sd[prefix + ".cls_token"].expand(b, -1, -1), x], dim=1)
    x = x + sd[prefix + ".pos_embed"]
    if prefix + ".sie_embed" in sd:
        x = x + sie_coef * sd[prefix + ".sie_embed"][cam]
    attns = []
    depth = 0
    while f"{prefix}.blocks.{depth}.norm1.weight" in sd:
        depth += 1
    for i in range(depth):
        keep, kp = None, 1.0
        if drop_keep is not None and drop_rates is not None and drop_rates[i] > 0:
            keep, kp = drop_keep[i].to(x.dtype), 1 - drop_rates[i]       # vit_pytorch.py:64: keep_prob = 1 - drop_prob (python floats)
        x, a = vit_block(x, sd, f"{prefix}.blocks.{i}", heads, 1e-6, keep, kp, qk_scale)
        attns.append(a)
    return _ln(x, sd, prefix + ".norm", 1e-6), attns


# ----------------------------------------------------------------------------------------------
# A4  attention rollout + per-head top-k (SFTS.py:145-164)
# ----------------------------------------------------------------------------------------------
def rollout_scores(attns):
    """CLS row of A_{L-1} @ ... @ A_0, patches only: (B, heads, N).  Matrix form, as the
    reference computes it (SFTS.py:150-153)."""
    m = attns[0]
    for a in attns[1:]:
        m = torch.matmul(a, m)
    return m[:, :, 0, 1:]


def part_attention_mask(scores, k):
    """OR over heads of per-head top-k masks (SFTS.py:154-162).  scores (B, heads, N) fp32."""
    b, h, n = scores.shape
    m = topk_mask(scores.reshape(b * h, n), k).reshape(b, h, n)
    return m.any(dim=1)


# ----------------------------------------------------------------------------------------------
# A5  SFTS mask application + background consistency loss (SFTS.py:181-230)
# ----------------------------------------------------------------------------------------------
def sfts_apply(feats, index, training):
    """feats: list of (B,T,D); index (B,N) bool.  Returns masked feats and loss_bg (or None)."""
    idx = index.unsqueeze(-1)
    out = [torch.cat([f[:, :1], f[:, 1:] * idx], dim=1) for f in feats]
    loss = None
    if training:
        bg = [f[:, 1:] * (~idx) for f in feats]
        loss = 0.0
        for i in range(len(bg)):
            for j in range(i + 1, len(bg)):
                loss = loss + F.mse_loss(bg[i], bg[j])
    return out, loss


# ----------------------------------------------------------------------------------------------
# A6  HMA head: masked blocks (vit_pytorch.py:309-352, 240-258, 158-168)
# ----------------------------------------------------------------------------------------------
def _masked_attention(x, mask, sd, p, heads):
    b, t, d = x.shape
    if t != mask.shape[1]:
        mask = mask.repeat(1, t // mask.shape[1], 1)
    x = x * mask
    hd = d // heads
    qkv = F.linear(x, sd[p + ".qkv.weight"]).reshape(b, t, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    m = mask.unsqueeze(1).repeat(1, heads, 1, 1)
    attn = attn.masked_fill((m @ m.transpose(-2, -1)) == 0, -65504.0)
    attn = attn.softmax(dim=-1) * m
    o = (attn @ v).transpose(1, 2).reshape(b, t, d)
    return F.linear(o, sd[p + ".proj.weight"])


def _masked_mlp(x, mask, sd, p):
    if x.shape[1] != mask.shape[1]:
        mask = mask.repeat(1, x.shape[1] // mask.shape[1], 1)
    x = x * mask
    return F.linear(F.gelu(F.linear(x, sd[p + ".fc1.weight"])), sd[p + ".fc2.weight"])


def hma_modality_blocks(feats, mask, sd, prefix="FUSE_block", heads=12, tags=("R", "N", "T")):
    """The three per-modality masked blocks (vit_pytorch.py:310-317). mask: (B,T,1) float."""
    out = []
    for f, tag in zip(feats, tags):
        f = f + _masked_attention(_ln(f, sd, f"{prefix}.norm{tag}", 1e-5), mask, sd,
                                  f"{prefix}.attn{tag}", heads)
        f = f + _masked_mlp(_ln(f, sd, f"{prefix}.norm{tag}_", 1e-5), mask, sd, f"{prefix}.mlp{tag}")
        out.append(f)
    return out


def hma_joint_block(feats, mask, sd, prefix="FUSE_block", heads=12):
    """Joint masked block on cat[R,N,T] + out_norm + re-mask (vit_pytorch.py:324-337)."""
    x = torch.cat(feats, dim=1)
    x = x + _masked_attention(_ln(x, sd, prefix + ".norm1", 1e-5), mask, sd, prefix + ".attn1", heads)
    x = x + _masked_mlp(_ln(x, sd, prefix + ".norm2", 1e-5), mask, sd, prefix + ".mlp")
    x = _ln(x, sd, prefix + ".out_norm", 1e-5)
    return x * mask.repeat(1, len(feats), 1)


# ----------------------------------------------------------------------------------------------
# A7  OCFR (OCFR.py:44-84,22-42) - functional: returns loss and the updated centre tables
# ----------------------------------------------------------------------------------------------
def ocfr(cls_feats, centers, label, momentum=0.8):
    """cls_feats: 3 x (B,D); centers: 3 x (C,D) (updated IN PLACE like the reference's
    Parameter assignment, OCFR.py:80-83); label (B,) int64 in P contiguous equal groups."""
    uniq = label.unique()
    mom = torch.tensor(momentum, dtype=torch.float32)          # OCFR.py:14 keeps it as an fp32 tensor
    chunk = label.shape[0] // uniq.shape[0]
    lab_first = label[::chunk]
    loss = 0.0
    for f, c in zip(cls_feats, centers):
        fn = F.normalize(f, dim=1)
        batch_c = torch.stack([fn[label == u].mean(dim=0) for u in uniq]).detach()
        with torch.no_grad():
            c[uniq] = mom * batch_c + (1 - mom) * c[uniq]
        sel = c[uniq]
        rows = torch.stack([sel[uniq == lab_first[i]].repeat(chunk, 1) for i in range(lab_first.shape[0])])
        loss = loss + F.mse_loss(rows.reshape(-1, f.shape[1]).detach(), fn)
    return loss


# ----------------------------------------------------------------------------------------------
# A8  EDITOR.forward glue (make_model.py:150-258)
# ----------------------------------------------------------------------------------------------
def _bn1d(x, sd, p, training, momentum=0.1, eps=1e-5):
    """nn.BatchNorm1d (make_model.py:115,120,140); running stats updated in place when training."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], training, momentum, eps)


# the reference's three modalities: (input key, name of the REDUCE layer / centre table, BlockMask tag)
MODALITIES3 = (("RGB", "RGB", "R"), ("NI", "NIR", "N"), ("TI", "TIR", "T"))
# 4-modal extension (BASELINE.json config 5; no counterpart in the reference, whose forward hard-codes three keys,
# make_model.py:153-155): the same computation with one more term in every per-modality loop / sum / OR / concat
MODALITIES4 = MODALITIES3 + (("M4", "M4", "M4"),)


def editor_forward(sd, x, cam, label=None, training=False, al=1, head_keep=2, frequency_keep=10,
                   heads=12, hma_heads=12, sie_coef=3.0, drop_keep=None, drop_rates=None,
                   teacher_index=None, return_aux=False, modalities=MODALITIES3, qk_scale=None):
    """EDITOR.forward (make_model.py:150-258).  `sd` maps state-dict names to tensors (leaf
    tensors requiring grad for a backward run; BN running stats / OCFR centres are mutated).
    teacher_index: optional (B,N) bool to force the SFTS selection (bf16 protocol).
    drop_keep: optional (nmod, depth, 2, B) 0/1 stochastic-depth keep masks with drop_rates (depth) - the backbone runs once
    per modality (make_model.py:158-160), every run draws its own masks, two per block (vit_pytorch.py:217-218)."""
    imgs = [x[m[0]] for m in modalities]
    nmod = len(imgs)
    aux = {}
    mask_fre, counts = frequency_mask(imgs[0], imgs[1], imgs[2] if nmod > 2 else None, frequency_keep, extra=imgs[3:])
    feats, masks, scores = [], [], []
    for img, dk in zip(imgs, range(nmod)):
        dkm = None if drop_keep is None else drop_keep[dk]
        f, attns = vit_forward(sd, img, cam, heads, sie_coef, drop_keep=dkm, drop_rates=drop_rates, qk_scale=qk_scale)
        n = f.shape[1] - 1
        k = int(n * ((1 / n) * int(head_keep)))                 # make_model.py:92, SFTS.py:155
        with torch.no_grad():
            sc = rollout_scores([a.detach() for a in attns])
        feats.append(f)
        scores.append(sc)
        masks.append(part_attention_mask(sc, k))
    index = mask_fre
    for mk in masks:
        index = index | mk
    if teacher_index is not None:
        index = teacher_index
    aux.update(mask_fre=mask_fre, counts=counts, attn_masks=masks, scores=scores, index=index)

    cls_tri = [f[:, 0] for f in feats]
    if training:
        if al:
            ori = torch.cat(cls_tri, dim=-1)
            ori_score = F.linear(_bn1d(ori, sd, "AL_BN", True), sd["AL_HEAD.weight"])
        else:
            mod_scores = [F.linear(_bn1d(c, sd, "BACKBONE_BN", True), sd["BACKBONE_HEAD.weight"])
                          for c in cls_tri]
    feats_s, loss_bcc = sfts_apply(feats, index, training)
    mask = torch.cat([torch.ones(index.shape[0], 1, 1), index.unsqueeze(-1).float()], dim=1)
    mods = hma_modality_blocks(feats_s, mask, sd, heads=hma_heads, tags=[m[2] for m in modalities])
    loss_ocfr = None
    if training:
        centers = [sd["FUSE_block.memory_cls.%s_centers" % m[1]] for m in modalities]
        loss_ocfr = ocfr([m[:, 0] for m in mods], centers, label)
    fused = hma_joint_block(mods, mask, sd, heads=hma_heads)
    t = feats[0].shape[1]
    parts = [fused[:, i * t:(i + 1) * t] for i in range(nmod)]
    num = (parts[0][:, 1:].sum(dim=2) != 0).sum(dim=1).unsqueeze(-1)   # RGB's count for all three
    red = []
    for part, tag in zip(parts, [m[1] for m in modalities]):
        pooled = part[:, 1:].sum(dim=1) / num
        red.append(F.linear(torch.cat([part[:, 0], pooled], dim=-1), sd[tag + "_REDUCE.weight"],
                            sd[tag + "_REDUCE.bias"]))
    cls4t = torch.cat(red, dim=-1)
    aux.update(num=num, loss_bcc=loss_bcc, loss_ocfr=loss_ocfr)
    if not training:
        return (cls4t, aux) if return_aux else cls4t
    score = F.linear(_bn1d(cls4t, sd, "FUSE_BN", True), sd["FUSE_HEAD.weight"])
    if al:
        out = (score, cls4t, ori_score, ori, loss_bcc + loss_ocfr)
    else:
        out = (score, cls4t) + tuple(v for pair in zip(mod_scores, cls_tri) for v in pair) + (loss_bcc + loss_ocfr,)
    return (out, aux) if return_aux else out


def projection_loss(outputs, seed=5):
    """Deterministic scalar objective used by the gradient fixtures (NOT the reference's loss,
    which is row N1 of SURVEY 8(f)): sum_i mean(out_i * R_i) + aux, with R_i seeded."""
    from editor_amd import synth
    total = outputs[-1]
    for i, o in enumerate(outputs[:-1]):
        r = synth.uniform(seed, "proj/%d" % i, tuple(o.shape))
        total = total + (o * r).mean()
    return total


# ----------------------------------------------------------------------------------------------
# N1  loss head (layers/make_loss.py:36-56; softmax_loss.py:4-34; triplet_loss.py:16-33,51-136)
# ----------------------------------------------------------------------------------------------
def cross_entropy_label_smooth(logits, target, eps=0.1):
    """CrossEntropyLabelSmooth.forward: (-((1-eps)*onehot + eps/K) * log_softmax).mean(0).sum()."""
    logp = F.log_softmax(logits, dim=1)
    k = logits.shape[1]
    soft = torch.zeros_like(logp).scatter_(1, target.unsqueeze(1), 1) * (1 - eps) + eps / k
    return (-soft * logp).mean(0).sum()


def triplet_soft_margin(feat, labels):
    """TripletLoss(margin=None): un-normalised Euclidean distances (clamp 1e-12, sqrt), batch-hard mining,
    SoftMarginLoss(dist_an - dist_ap, 1)."""
    n = feat.shape[0]
    sq = feat.pow(2).sum(1, keepdim=True)
    dist = (sq.expand(n, n) + sq.expand(n, n).t() - 2 * feat @ feat.t()).clamp(min=1e-12).sqrt()
    same = labels.view(n, 1).eq(labels.view(1, n))
    d_ap = dist[same].view(n, -1).max(1).values
    d_an = dist[~same].view(n, -1).min(1).values
    return F.soft_margin_loss(d_an - d_ap, torch.ones_like(d_an))


def center_loss(x, centers, labels):
    """CenterLoss.forward (layers/center_loss.py:30-51): expanded squared distances, masked to the own class, EVERY entry of the
    (B, C) matrix clamped to [1e-12, 1e12] (the masked-out zeros become 1e-12 each), summed, / B."""
    b, c = x.shape[0], centers.shape[0]
    distmat = x.pow(2).sum(dim=1, keepdim=True).expand(b, c) + centers.pow(2).sum(dim=1, keepdim=True).expand(c, b).t()
    distmat = distmat - 2.0 * (x @ centers.t())
    mask = labels.unsqueeze(1).expand(b, c).eq(torch.arange(c).long().expand(b, c))
    return (distmat * mask.float()).clamp(min=1e-12, max=1e12).sum() / b


def loss_pairs(output, target):
    """engine/processor.py:82-92: (score_i, feat_i) pairs + trailing aux loss."""
    loss = output[-1] if len(output) % 2 == 1 else 0.0
    for i in range(0, len(output) - (len(output) % 2), 2):
        loss = loss + cross_entropy_label_smooth(output[i], target) + triplet_soft_margin(output[i + 1], target)
    return loss
