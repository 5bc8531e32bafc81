"""Optimizer wiring of the reference's engines (engine/forgery_engine.py:149-156): weight-decay parameter
groups as timm's ``param_groups_weight_decay`` builds them, AdamW(amsgrad) and StepLR from the YAML keys.
The optimizer itself stays torch's (host-side torch AdamW is acceptable; a fused
multi-tensor HIP AdamW is row (f)2)."""
import torch


def param_groups_weight_decay(model, weight_decay=1e-5, no_weight_decay_list=()):
    """No decay for 1-D / scalar tensors, '.bias' and listed names; frozen parameters are skipped
    (timm 0.9.16 semantics)."""
    no_weight_decay_list = set(no_weight_decay_list)
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if p.ndim <= 1 or name.endswith(".bias") or name in no_weight_decay_list:
            no_decay.append(p)
        else:
            decay.append(p)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


OPTIMIZERS = {"adamw": torch.optim.AdamW, "adam": torch.optim.Adam, "sgd": torch.optim.SGD}


def get_optimizer(name, params, **kwargs):
    return OPTIMIZERS[name.lower()](params, **kwargs)


def build_optimizer(model, opt_cfg):
    """opt_cfg: the YAML's config.optimizer dict (name, weight_decay, lr, betas, amsgrad, ...)."""
    cfg = dict(opt_cfg)
    name = cfg.pop("name")
    wd = cfg.pop("weight_decay", 0.0)
    if "betas" in cfg:
        cfg["betas"] = tuple(cfg["betas"])
    groups = param_groups_weight_decay(model, wd)
    # torch's fused multi-tensor Adam(W) streams the 128 M parameters + 3 state tensors at 3.8 TB/s on MI355X
    # (1.2 ms per step vs 3.1 ms for the foreach implementation; tools/bench_optim.py): SURVEY row (f)2 needs no
    # custom kernel beyond it
    if name.lower() in ("adamw", "adam") and "fused" not in cfg and "foreach" not in cfg \
            and all(p.is_cuda for g in groups for p in g["params"]):
        cfg["fused"] = True
    return get_optimizer(name, groups, **cfg)


class ConstantLR(torch.optim.lr_scheduler.LRScheduler):
    """scheduler/__init__.py:13-18 of the reference: what an absent `scheduler:` block means."""

    def get_lr(self):
        return list(self.base_lrs)


SCHEDULERS = {
    "ConstantLR": ConstantLR,
    "StepLR": torch.optim.lr_scheduler.StepLR,
    "MultiStepLR": torch.optim.lr_scheduler.MultiStepLR,
    "ExponentialLR": torch.optim.lr_scheduler.ExponentialLR,
    "CosineAnnealingLR": torch.optim.lr_scheduler.CosineAnnealingLR,
    "CosineAnnealingWarmRestarts": torch.optim.lr_scheduler.CosineAnnealingWarmRestarts,
}


def build_scheduler(optimizer, sched_cfg):
    """scheduler.get_scheduler (scheduler/__init__.py:33-40): the YAML's config.scheduler dict (name + kwargs) or None.
    The timm schedulers of the reference's table (TimmStepLR, TimmCosineLR) are not available in this image."""
    if sched_cfg is None:
        return ConstantLR(optimizer)
    cfg = dict(sched_cfg)
    name = cfg.pop("name")
    if name not in SCHEDULERS:
        raise KeyError(f"scheduler '{name}' is not available here; known: {sorted(SCHEDULERS)}")
    return SCHEDULERS[name](optimizer, **cfg)
