"""Loss head that CONSUMES the hot path's train-mode outputs - row N1 of SURVEY.md 8(f).

Mirrors the reference's loss assembly so that bench.py times a complete training step and a harness written for
engine/processor.py:82-92 runs unchanged:
    CrossEntropyLabelSmooth(eps=0.1)                                   layers/softmax_loss.py:4-34
    TripletLoss() soft-margin, batch-hard, un-normalised Euclidean     layers/triplet_loss.py:51-136
    loss_func = ID_LOSS_WEIGHT * xent + TRIPLET_LOSS_WEIGHT * triplet  layers/make_loss.py:36-56
Both terms run as HIP kernels (csrc/loss.hip) through the C ABI; CPU tensors are rejected, there is no fallback.
"""
import torch

from . import functional as Fn


def cross_entropy_label_smooth(logits, target, eps=0.1):
    return Fn.CrossEntropyLabelSmoothFn.apply(logits, target, float(eps))


def triplet_soft_margin(feat, labels):
    return Fn.TripletSoftMarginFn.apply(feat, labels)


def make_loss(cfg=None, num_classes=None):
    """layers/make_loss.py:13-56 for the configuration the reference trains with (sampler softmax_triplet,
    METRIC_LOSS_TYPE triplet, NO_MARGIN, label smoothing on).  Returns loss_func(score, feat, target, target_cam)."""
    idw = float(getattr(getattr(cfg, "MODEL", None), "ID_LOSS_WEIGHT", 1.0)) if cfg is not None else 1.0
    trw = float(getattr(getattr(cfg, "MODEL", None), "TRIPLET_LOSS_WEIGHT", 1.0)) if cfg is not None else 1.0

    def loss_func(score, feat, target, target_cam=None):
        return idw * cross_entropy_label_smooth(score, target) + trw * triplet_soft_margin(feat, target)

    return loss_func


def loss_pairs(output, target, loss_fn=None):
    """engine/processor.py:82-92: odd-length output = (score_i, feat_i) pairs + trailing aux loss."""
    loss_fn = loss_fn or make_loss()
    npair = len(output) - (len(output) % 2)
    loss = output[-1] if len(output) % 2 == 1 else None
    for i in range(0, npair, 2):
        term = loss_fn(score=output[i], feat=output[i + 1], target=target)
        loss = term if loss is None else loss + term
    return loss
