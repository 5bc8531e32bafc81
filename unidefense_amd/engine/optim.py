"""Optimizer wiring of the reference's engines (engine/forgery_engine.py:149-156): weight-decay parameter
groups as timm's ``param_groups_weight_decay`` builds them, AdamW(amsgrad) and StepLR from the YAML keys.
AdamW on the GPU is the multi-tensor HIP kernel of csrc/optim.hip; other optimizers and CPU
parameters use torch's."""
import math

import torch

from ..config import cfg as _cfg


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


class HipAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, optional amsgrad, param groups) with the whole parameter set
    updated by ONE launch of the multi-tensor HIP kernel (csrc/optim.hip: ud_adamw_multi): per-group lr / weight
    decay, bias corrections from a device-side step counter, GradScaler's 1/scale and found_inf skip read on the device
    (`_step_supports_amp_scaling`: GradScaler.step hands both over instead of unscaling in a pass of its own).
    Reference: engine/forgery_engine.py:149-156,228; engine/abstract_engine.py:281-283,374-378.
    State layout: exp_avg / exp_avg_sq / max_exp_avg_sq are views of three flat buffers (16-byte aligned per tensor);
    state_dict() / load_state_dict() speak torch.optim.AdamW's format (incl. the per-parameter `step`), so a checkpointed
    optimizer resumes with its moments and bias corrections, and either optimizer can load the other's state.
    ONE step counter serves every parameter (the kernel derives the bias corrections from it): state_dict() stamps it on
    every entry, load_state_dict() requires the loaded per-parameter steps to agree — which they do whenever every parameter
    received a gradient in every step, as in the reference's engines; a state in which they differ (parameters frozen for part
    of a torch.optim.AdamW run) is refused rather than silently averaged."""

    _step_supports_amp_scaling = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, maximize=False):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, maximize=maximize)
        super().__init__(params, defaults)
        if len(self.param_groups) > 8:
            raise ValueError("HipAdamW supports up to 8 parameter groups")
        g0 = self.param_groups[0]
        for g in self.param_groups:
            if (g["betas"], g["eps"], g["amsgrad"], g["maximize"]) != (g0["betas"], g0["eps"], g0["amsgrad"], g0["maximize"]):
                raise ValueError("HipAdamW: betas / eps / amsgrad / maximize must agree across groups")
        self._plan = None
        self._steps = None          # device int32 [2]: ping-pong step counter (a skipped step does not advance it)
        self._cur = 0
        self._loaded_step = 0       # applied steps of a restored state (load_state_dict)

    def _alloc_state(self, dev):
        import ctypes as C
        from .. import lib
        chunk = lib.call("ud_adamw_chunk_elems")
        ams = self.param_groups[0]["amsgrad"]
        total, offs = 0, {}
        for g in self.param_groups:
            for p in g["params"]:
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise ValueError("HipAdamW updates contiguous fp32 CUDA parameters")
                offs[p] = total
                total += (p.numel() + 3) // 4 * 4
        names = ["exp_avg", "exp_avg_sq"] + (["max_exp_avg_sq"] if ams else [])
        flat = {k: torch.zeros(total, dtype=torch.float32, device=dev) for k in names}
        for p, off in offs.items():
            st = self.state[p]
            for k in names:
                view = flat[k][off:off + p.numel()].view_as(p)
                old = st.get(k)
                if old is not None:                    # restored by load_state_dict (ours or torch.optim.AdamW's): keep it
                    view.copy_(old.to(device=dev, dtype=torch.float32).view_as(p))
                st[k] = view
        self._flat, self._chunk = flat, chunk
        # applied steps so far (bias corrections continue where a restored run stopped: load_state_dict)
        self._steps = torch.full((2,), int(self._loaded_step), dtype=torch.int32, device=dev)
        self._cur = 0

    def _plan_stale(self, active):
        """The cached parameter / state pointers no longer describe the tensors (load_state_dict replaced the state,
        the parameters were moved): checked on the first and last tensor, every step."""
        st = self._plan["static"]
        for i in (0, len(active) - 1):
            p = active[i][0]
            if st[i][0] != p.data_ptr() or st[i][1] != self.state[p]["exp_avg"].data_ptr():
                return True
        return False

    def state_dict(self):
        """torch.optim.AdamW's layout: per parameter `step` (applied steps, one device read here), exp_avg, exp_avg_sq
        [, max_exp_avg_sq] — interchangeable with torch's optimizer in both directions."""
        if self._steps is not None:
            n = float(self.step_count())
            for st in self.state.values():
                st["step"] = torch.tensor(n)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """The restored moments are copied into this optimizer's flat buffers and the step counter is set HERE, not at the
        next step(): torch hands state tensors over by reference when dtype and device already match, and a source
        optimizer that is still alive would go on updating them."""
        super().load_state_dict(state_dict)
        self._plan = None
        self._steps = None
        steps = {int(float(st["step"])) for st in self.state.values() if "step" in st}
        if len(steps) > 1:
            raise ValueError(f"HipAdamW keeps one step counter for all parameters; the loaded state has {sorted(steps)}")
        self._loaded_step = max(steps) if steps else 0
        params = [p for g in self.param_groups for p in g["params"]]
        if params and all(p.is_cuda for p in params):
            self._alloc_state(params[0].device)

    def _make_plan(self, active, dev):
        """Static per set of updated tensors: chunk map and pointer table on the device."""
        rows = []
        for ti, (p, gi) in enumerate(active):
            rows.extend((ti, c) for c in range(-(-p.numel() // self._chunk)))
        chunk_map = torch.tensor(rows, dtype=torch.int32).contiguous().to(dev)
        ams = self.param_groups[0]["amsgrad"]
        static = []                                    # everything of a table row but the gradient pointer
        for p, gi in active:
            st = self.state[p]
            static.append((p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                           st["max_exp_avg_sq"].data_ptr() if ams else 0, p.numel(), gi))
        # tables: one device pointer table per distinct SET of gradient buffers (the engine's two captured passes hand
        # over two fixed sets, an eager backward a new one every time) — a steady-state step uploads nothing
        return {"key": tuple(id(p) for p, _ in active), "chunk_map": chunk_map, "n_chunks": len(rows), "static": static,
                "tables": {}}

    @torch.no_grad()
    def step(self, closure=None):
        import ctypes as C
        from .. import lib
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        active = [(p, gi) for gi, g in enumerate(self.param_groups) for p in g["params"] if p.grad is not None]
        if not active:
            return loss
        dev = active[0][0].device
        if self._steps is None:
            self._alloc_state(dev)
        if self._plan is None or self._plan["key"] != tuple(id(p) for p, _ in active) or self._plan_stale(active):
            self._plan = self._make_plan(active, dev)
        plan = self._plan
        ams = self.param_groups[0]["amsgrad"]
        gptrs = []
        for p, gi in active:
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = p.grad = g.contiguous().to(torch.float32)
            gptrs.append(g.data_ptr())
        gkey = tuple(gptrs)
        table = plan["tables"].get(gkey)
        if table is None:                             # new set of gradient buffers: upload 64 B per tensor (32 KB)
            rows = [(st[0], gp, st[1], st[2], st[3], st[4], st[5], 0) for st, gp in zip(plan["static"], gptrs)]
            if len(plan["tables"]) >= 8:               # eager backwards allocate new buffers every step: keep it bounded
                plan["tables"].clear()
            table = plan["tables"][gkey] = torch.tensor(rows, dtype=torch.int64).to(dev)
        ng = len(self.param_groups)
        lr = (C.c_float * ng)(*[float(g["lr"]) for g in self.param_groups])
        wd = (C.c_float * ng)(*[float(g["weight_decay"]) for g in self.param_groups])
        g0 = self.param_groups[0]
        grad_scale, found_inf = getattr(self, "grad_scale", None), getattr(self, "found_inf", None)

        def dp(t):
            return None if t is None else C.c_void_p(t.data_ptr())
        if found_inf is not None and found_inf.dtype != torch.float32:
            found_inf = found_inf.float()
        if grad_scale is not None and grad_scale.dtype != torch.float32:
            grad_scale = grad_scale.float()
        s_in = C.c_void_p(self._steps.data_ptr() + 4 * self._cur)
        s_out = C.c_void_p(self._steps.data_ptr() + 4 * (1 - self._cur))
        lib.call("ud_adamw_multi", dp(table), dp(plan["chunk_map"]), plan["n_chunks"], lr, wd, ng,
                 float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"]), int(bool(ams)), int(bool(g0["maximize"])),
                 dp(grad_scale), dp(found_inf), s_in, s_out, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        self._cur = 1 - self._cur
        # the kernel wrote the parameters through raw pointers: tell autograd (and anything keyed on Tensor._version)
        torch._C._increment_version([p_ for group in self.param_groups for p_ in group["params"] if p_.grad is not None])
        return loss

    def step_count(self):
        """Number of applied (not skipped) steps — one device read."""
        return 0 if self._steps is None else int(self._steps[self._cur].item())


OPTIMIZERS = {"adamw": torch.optim.AdamW, "adam": torch.optim.Adam, "sgd": torch.optim.SGD, "hip_adamw": HipAdamW}


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
    on_gpu = all(p.is_cuda for g in groups for p in g["params"])
    # AdamW on the GPU = the multi-tensor HIP kernel (one launch for the 504 tensors, GradScaler's
    # unscale / found_inf folded in); cfg.hip_adamw = False or explicit fused / foreach keys keep torch's implementation
    if name.lower() == "adamw" and on_gpu and "fused" not in cfg and "foreach" not in cfg \
            and _cfg.hip_adamw:
        return HipAdamW(groups, **cfg)
    if name.lower() in ("adamw", "adam") and "fused" not in cfg and "foreach" not in cfg and on_gpu:
        cfg["fused"] = True
    return get_optimizer(name, groups, **cfg)


class ConstantLR(torch.optim.lr_scheduler.LRScheduler):
    """scheduler/__init__.py:13-18 of the reference: what an absent `scheduler:` block means."""

    def get_lr(self):
        return list(self.base_lrs)


class _TimmScheduler:
    """Base of the two timm schedulers in the reference's table (scheduler/__init__.py:9-10,24,31; timm 0.9.16 per README.md:64,
    THIRD-PARTY SOURCE ABSENT from this image: the published algorithm of timm/scheduler/{scheduler,step_lr,cosine_lr}.py is
    restated here, parity unpinned).  timm's `step(epoch)` wants the epoch; the reference's engines call `scheduler.step()` with
    no argument (engine/abstract_engine.py:203,378 — a TypeError on the real timm class) and read `get_last_lr()`
    (uniattack_engine.py:347), so `step()` without an argument counts its own calls and `get_last_lr()` exists.  The lr-noise
    options are not restated (no shipped YAML uses these schedulers at all)."""

    def __init__(self, optimizer, warmup_t=0, warmup_lr_init=0.0, warmup_prefix=False, t_in_epochs=True, initialize=True, **kw):
        if kw:
            raise TypeError(f"{type(self).__name__}: unsupported arguments {sorted(kw)} (the lr-noise options of timm are not restated)")
        self.optimizer, self.t_in_epochs = optimizer, t_in_epochs
        for g in optimizer.param_groups:
            if initialize:
                g.setdefault("initial_lr", g["lr"])
            elif "initial_lr" not in g:
                raise KeyError("initial_lr is not specified in a param_group of the optimizer")
        self.base_values = [g["initial_lr"] for g in optimizer.param_groups]
        self.warmup_t, self.warmup_lr_init, self.warmup_prefix = warmup_t, warmup_lr_init, warmup_prefix
        self._t = 0
        if warmup_t:
            self.warmup_steps = [(v - warmup_lr_init) / warmup_t for v in self.base_values]
            self._set([warmup_lr_init] * len(self.base_values))
        else:
            self.warmup_steps = [1.0 for _ in self.base_values]

    def _set(self, values):
        for g, v in zip(self.optimizer.param_groups, values):
            g["lr"] = v

    def _lr_at(self, t):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * s for s in self.warmup_steps]
        return self._decayed(t - self.warmup_t if self.warmup_prefix else t)

    def step(self, epoch=None, metric=None):
        if epoch is None:
            self._t += 1
            epoch = self._t
        else:
            self._t = epoch
        if self.t_in_epochs:
            self._set(self._lr_at(epoch))

    def step_update(self, num_updates, metric=None):
        if not self.t_in_epochs:
            self._set(self._lr_at(num_updates))

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, state):
        self.__dict__.update(state)


class TimmStepLR(_TimmScheduler):
    """timm.scheduler.StepLRScheduler: lr = base * decay_rate ** (t // decay_t) after a linear warm-up of warmup_t steps from
    warmup_lr_init (warmup_prefix defaults to True there: the decay clock starts when the warm-up ends)."""

    def __init__(self, optimizer, decay_t, decay_rate=1.0, warmup_t=0, warmup_lr_init=0.0, warmup_prefix=True, **kw):
        self.decay_t, self.decay_rate = decay_t, decay_rate
        super().__init__(optimizer, warmup_t, warmup_lr_init, warmup_prefix, **kw)

    def _decayed(self, t):
        return [v * (self.decay_rate ** (t // self.decay_t)) for v in self.base_values]


class TimmCosineLR(_TimmScheduler):
    """timm.scheduler.CosineLRScheduler (SGDR with cycle_mul / cycle_decay / cycle_limit and the k-decay exponent):
    lr = lr_min + (lr_max * cycle_decay**i - lr_min) / 2 * (1 + cos(pi * t_curr**k / t_i**k)) inside cycle i < cycle_limit,
    lr_min afterwards."""

    def __init__(self, optimizer, t_initial, lr_min=0.0, cycle_mul=1.0, cycle_decay=1.0, cycle_limit=1, warmup_t=0,
                 warmup_lr_init=0.0, warmup_prefix=False, k_decay=1.0, **kw):
        assert t_initial > 0 and lr_min >= 0
        self.t_initial, self.lr_min, self.cycle_mul, self.cycle_decay = t_initial, lr_min, cycle_mul, cycle_decay
        self.cycle_limit, self.k_decay = cycle_limit, k_decay
        super().__init__(optimizer, warmup_t, warmup_lr_init, warmup_prefix, **kw)

    def _decayed(self, t):
        if self.cycle_mul != 1:
            i = math.floor(math.log(1 - t / self.t_initial * (1 - self.cycle_mul), self.cycle_mul))
            t_i = self.cycle_mul ** i * self.t_initial
            t_curr = t - (1 - self.cycle_mul ** i) / (1 - self.cycle_mul) * self.t_initial
        else:
            i = t // self.t_initial
            t_i = self.t_initial
            t_curr = t - self.t_initial * i
        if i >= self.cycle_limit:
            return [self.lr_min for _ in self.base_values]
        gamma, k = self.cycle_decay ** i, self.k_decay
        return [self.lr_min + 0.5 * (v * gamma - self.lr_min) * (1 + math.cos(math.pi * t_curr ** k / t_i ** k))
                for v in self.base_values]


SCHEDULERS = {
    "ConstantLR": ConstantLR,
    "StepLR": torch.optim.lr_scheduler.StepLR,
    "TimmStepLR": TimmStepLR,
    "MultiStepLR": torch.optim.lr_scheduler.MultiStepLR,
    "ExponentialLR": torch.optim.lr_scheduler.ExponentialLR,
    "CosineAnnealingLR": torch.optim.lr_scheduler.CosineAnnealingLR,
    "CosineAnnealingWarmRestarts": torch.optim.lr_scheduler.CosineAnnealingWarmRestarts,
    "ReduceLROnPlateau": torch.optim.lr_scheduler.ReduceLROnPlateau,
    "TimmCosineLR": TimmCosineLR,
}


def build_scheduler(optimizer, sched_cfg):
    """scheduler.get_scheduler (scheduler/__init__.py:33-40): the YAML's config.scheduler dict (name + kwargs) or None; the
    reference's nine names."""
    if sched_cfg is None:
        return ConstantLR(optimizer)
    cfg = dict(sched_cfg)
    name = cfg.pop("name")
    if name not in SCHEDULERS:
        raise KeyError(f"scheduler '{name}' is not one of the reference's: {sorted(SCHEDULERS)}")
    return SCHEDULERS[name](optimizer, **cfg)
