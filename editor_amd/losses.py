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


class CenterLoss(torch.nn.Module):
    """layers/center_loss.py:9-51: `centers` (C, feat_dim) ~ N(0,1) and forward(x, labels) = sum of the masked (B, C) squared-distance
    matrix, every entry clamped to [1e-12, 1e12], / B - as HIP kernels (csrc/loss.hip: editor_center_loss_fwd / _bwd).  make_loss builds
    it with use_gpu=False (the parameter starts on the CPU, train_net.py:72-74 hands it to make_optimizer); with the shipped
    METRIC_LOSS_TYPE ('triplet') its forward is never called (layers/make_loss.py:36-56 has no centre term; processor.py:97-100 only
    rescales its gradient when 'center' is in the loss type) - it is here so that a harness which does call it runs.  Like every op of
    this package it needs GPU tensors: move the module with .cuda() / .to(device) first."""

    def __init__(self, num_classes=751, feat_dim=2048, use_gpu=False):
        super().__init__()
        self.num_classes, self.feat_dim, self.use_gpu = num_classes, feat_dim, use_gpu
        c = torch.randn(num_classes, feat_dim)
        self.centers = torch.nn.Parameter(c.cuda() if use_gpu else c)

    def forward(self, x, labels):
        assert x.size(0) == labels.size(0), "features.size(0) is not equal to labels.size(0)"
        if not (x.is_cuda and self.centers.is_cuda):
            raise RuntimeError("CenterLoss (MI355X build): features and centers must be on the GPU; there is no CPU fallback path")
        return Fn.CenterLossFn.apply(x, self.centers, labels)


def _loss_func(cfg=None):
    idw = float(getattr(getattr(cfg, "MODEL", None), "ID_LOSS_WEIGHT", 1.0)) if cfg is not None else 1.0
    trw = float(getattr(getattr(cfg, "MODEL", None), "TRIPLET_LOSS_WEIGHT", 1.0)) if cfg is not None else 1.0

    def loss_func(score, feat, target, target_cam=None):
        if feat.shape[0] != target.shape[0]:                          # make_loss.py:38-39
            target = target.repeat(feat.shape[0] // target.shape[0])
        # (a weight of exactly 1.0 - the shipped value of both - is not multiplied in: x * 1.0 == x bit for bit, and each product is
        #  a one-element launch in the forward and another in the backward)
        ce, tri = cross_entropy_label_smooth(score, target), triplet_soft_margin(feat, target)
        return (ce if idw == 1.0 else idw * ce) + (tri if trw == 1.0 else trw * tri)

    return loss_func


def make_loss(cfg=None, num_classes=None):
    """layers/make_loss.py:13-80 for the configuration the reference trains with (sampler softmax_triplet,
    METRIC_LOSS_TYPE triplet, NO_MARGIN, label smoothing on).  Returns (loss_func(score, feat, target, target_cam),
    center_criterion) as the reference does (train_net.py:72 unpacks both)."""
    return _loss_func(cfg), CenterLoss(num_classes=num_classes or 751, feat_dim=2048)


def loss_pairs(output, target, loss_fn=None):
    """engine/processor.py:82-92: odd-length output = (score_i, feat_i) pairs + trailing aux loss."""
    loss_fn = loss_fn or _loss_func()
    npair = len(output) - (len(output) % 2)
    loss = output[-1] if len(output) % 2 == 1 else None
    for i in range(0, npair, 2):
        term = loss_fn(score=output[i], feat=output[i + 1], target=target)
        loss = term if loss is None else loss + term
    return loss
