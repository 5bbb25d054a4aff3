"""Optimizer factory and learning-rate schedule of the training loop around the hot path (row N4).

Same names and call contracts as the reference's `solver` package, so `train_net.py:74-77` reads unchanged:

    optimizer, optimizer_center = make_optimizer(cfg, model, center_criterion)      solver/make_optimizer.py:4-29
    scheduler = create_scheduler(cfg, optimizer)                                    solver/scheduler_factory.py:7-31
    scheduler.step(epoch)                                                           engine/processor.py:68

`make_optimizer` returns editor_amd.optim.FusedSGD (one HIP launch per step) for OPTIMIZER_NAME == 'SGD'.  The
schedule is the reference's configuration of its CosineLRScheduler (solver/cosine_lr.py:67-94): linear warm-up from
0.01*BASE_LR over WARMUP_ITERS epochs, then ONE cosine cycle of MAX_EPOCHS epochs down to 0.001*BASE_LR, then the
floor; evaluated in closed form per epoch (host arithmetic on a handful of floats) and pushed into the optimizer's
device-resident lr table, which is what keeps a captured hipGraph step valid across epochs."""
import math

import torch

from .optim import FusedAdamW, FusedSGD


def param_group_table(cfg, names):
    """(name, lr, weight_decay) of every trainable parameter name - the group rule of solver/make_optimizer.py:6-19:
    "bias" in the NAME -> BASE_LR * BIAS_LR_FACTOR and WEIGHT_DECAY_BIAS; LARGE_FC_LR doubles "classifier"/"arcface"."""
    s = cfg.SOLVER
    table = []
    for n in names:
        lr, wd = s.BASE_LR, s.WEIGHT_DECAY
        if "bias" in n:
            lr, wd = s.BASE_LR * s.BIAS_LR_FACTOR, s.WEIGHT_DECAY_BIAS
        if getattr(s, "LARGE_FC_LR", False) and ("classifier" in n or "arcface" in n):
            lr = s.BASE_LR * 2
        table.append((n, lr, wd))
    return table


def make_optimizer(cfg, model, center_criterion=None):
    """solver/make_optimizer.py:4-29 -> (FusedSGD over the groups of param_group_table, SGD of the centre criterion)."""
    s = cfg.SOLVER
    name = getattr(s, "OPTIMIZER_NAME", "SGD")
    common = dict(shadow_dtype=_shadow_dtype(model), split_pairs=bool(getattr(getattr(model, "module", model), "split_fwd", False)))
    if name == "SGD":
        opt = FusedSGD(model.named_parameters(), base_lr=s.BASE_LR, weight_decay=s.WEIGHT_DECAY,
                       bias_lr_factor=s.BIAS_LR_FACTOR, weight_decay_bias=s.WEIGHT_DECAY_BIAS, momentum=s.MOMENTUM, **common)
    elif name == "AdamW":
        # make_optimizer.py:23-24: torch.optim.AdamW(params, lr=BASE_LR, weight_decay=WEIGHT_DECAY) - the per-parameter groups built
        # above it (:6-19) carry their own lr / weight decay, which override those defaults; betas / eps are torch's
        opt = FusedAdamW(model.named_parameters(), base_lr=s.BASE_LR, weight_decay=s.WEIGHT_DECAY,
                         bias_lr_factor=s.BIAS_LR_FACTOR, weight_decay_bias=s.WEIGHT_DECAY_BIAS, **common)
    else:
        raise NotImplementedError("the fused HIP update implements the optimizers the reference names (solver/make_optimizer.py:21-24: "
                                  "'SGD', 'AdamW'); got %r" % (name,))
    for g, (n, lr, wd) in zip(opt.param_groups, param_group_table(cfg, [g["name"] for g in opt.param_groups])):
        g["lr"], g["weight_decay"] = lr, wd
    opt.sync_param_groups()
    opt_center = None
    if center_criterion is not None:
        opt_center = torch.optim.SGD(center_criterion.parameters(), lr=getattr(s, "CENTER_LR", 0.5))
    return opt, opt_center


def _shadow_dtype(model):
    m = getattr(model, "module", model)
    dt = getattr(m, "act_dtype", torch.bfloat16)
    return None if dt == torch.float32 else dt


class WarmupCosineSchedule:
    """lr(epoch) for every parameter group.  Semantics of CosineLRScheduler as create_scheduler configures it
    (t_mul = 1, cycle_limit = 1, t_in_epochs, no noise): base values are the groups' lr at construction; the
    constructor already writes the warm-up start value into the groups (cosine_lr.py:62-64)."""

    def __init__(self, optimizer, t_initial, lr_min=0.0, decay_rate=1.0, warmup_t=0, warmup_lr_init=0.0, cycle_limit=1):
        if t_initial <= 0 or lr_min < 0:
            raise ValueError("t_initial > 0 and lr_min >= 0 required")
        self.optimizer = optimizer
        for g in optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
        self.base_values = [g["initial_lr"] for g in optimizer.param_groups]
        self.t_initial, self.lr_min, self.decay_rate = int(t_initial), float(lr_min), float(decay_rate)
        self.warmup_t, self.warmup_lr_init, self.cycle_limit = int(warmup_t), float(warmup_lr_init), int(cycle_limit)
        self._write([self.warmup_lr_init] * len(self.base_values) if self.warmup_t else self.base_values)

    def _get_lr(self, t):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * ((v - self.warmup_lr_init) / self.warmup_t) for v in self.base_values]
        cycle, t_curr = divmod(t, self.t_initial)
        if self.cycle_limit and cycle >= self.cycle_limit:
            return [self.lr_min for _ in self.base_values]
        gamma = self.decay_rate ** cycle
        floor = self.lr_min * gamma
        wave = 1 + math.cos(math.pi * t_curr / self.t_initial)
        return [floor + 0.5 * (v * gamma - floor) * wave for v in self.base_values]

    def get_epoch_values(self, epoch):
        return self._get_lr(epoch)

    def step(self, epoch, metric=None):
        self._write(self._get_lr(epoch))

    def step_update(self, num_updates, metric=None):         # the reference's schedule is per epoch (t_in_epochs)
        return None

    def _write(self, values):
        for g, v in zip(self.optimizer.param_groups, values):
            g["lr"] = v
        sync = getattr(self.optimizer, "sync_param_groups", None)
        if sync is not None:
            sync()

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, sd):
        self.__dict__.update(sd)


CosineLRScheduler = WarmupCosineSchedule          # the reference's class name (solver/cosine_lr.py:18)


def create_scheduler(cfg, optimizer):
    """solver/scheduler_factory.py:7-31."""
    s = cfg.SOLVER
    return WarmupCosineSchedule(optimizer, t_initial=s.MAX_EPOCHS, lr_min=0.001 * s.BASE_LR, decay_rate=0.1,
                                warmup_lr_init=0.01 * s.BASE_LR, warmup_t=s.WARMUP_ITERS, cycle_limit=1)
