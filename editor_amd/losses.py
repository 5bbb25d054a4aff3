"""Loss head that CONSUMES the hot path's outputs - row N1 of SURVEY.md 8(f) ("next", not yet a HIP kernel).

Host-side tensor ops (device plumbing) restating the reference's loss assembly so that bench.py times a
complete training step and the harness matches engine/processor.py:82-92:
    CrossEntropyLabelSmooth(eps=0.1)   layers/softmax_loss.py:4-34
    TripletLoss() soft-margin, batch-hard, un-normalised Euclidean   layers/triplet_loss.py:51-136
    loss_func = ID_LOSS_WEIGHT * xent + TRIPLET_LOSS_WEIGHT * triplet   layers/make_loss.py:36-56
"""
import torch
import torch.nn.functional as F


def cross_entropy_label_smooth(logits, target, eps=0.1):
    logp = F.log_softmax(logits.float(), dim=1)
    c = logits.shape[1]
    onehot = torch.zeros_like(logp).scatter_(1, target.unsqueeze(1), 1)
    soft = (1 - eps) * onehot + eps / c
    return (-soft * logp).mean(0).sum()


def triplet_soft_margin(feat, labels):
    feat = feat.float()
    n = feat.shape[0]
    sq = feat.pow(2).sum(1, keepdim=True)
    dist = (sq + sq.t() - 2 * feat @ feat.t()).clamp(min=1e-12).sqrt()
    same = labels.view(n, 1).eq(labels.view(1, n))
    d_ap = dist.masked_fill(~same, float("-inf")).max(1).values
    d_an = dist.masked_fill(same, float("inf")).min(1).values
    return F.soft_margin_loss(d_an - d_ap, torch.ones_like(d_an))


def loss_pairs(output, target):
    """engine/processor.py:82-92: odd-length output = (score_i, feat_i) pairs + trailing aux loss."""
    loss = output[-1]
    for i in range(0, len(output) - 1, 2):
        loss = loss + cross_entropy_label_smooth(output[i], target) + triplet_soft_margin(output[i + 1], target)
    return loss
