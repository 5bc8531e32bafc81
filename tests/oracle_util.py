"""Helpers shared by the tests: build oracle state, run oracle fwd/bwd with seeded masks."""
import os

import torch

from oracle import eb4, losses, param_fill

def job_cores():
    """CPU cores this process may really use: min(affinity mask, cgroup-v2 cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def fit_cpu_threads():
    """The CPU oracle on the cores of the JOB.  The GPU boxes show 256 logical CPUs under a 16-CPU quota and torch starts 128
    threads there: the float64 oracle of one element-wise gradient test then takes 51 s instead of 3.4 s (tools/
    probe_oracle_threads.py) — two thirds of the GPU suite's wall time were throttled OpenMP threads.  Children inherit the count
    through OMP_NUM_THREADS."""
    n = job_cores()
    os.environ.setdefault("OMP_NUM_THREADS", str(n))
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return n


LAMBDAS = dict(lambda_triplet=0.1, lambda_recons=0.1, lambda_freq=1.0, lambda_mask=0.1, lambda_fac=0.1)


def make_rng(n, seed, drop_rate, nblk=32, dc_rate=0.2):
    """Same construction as oracle/make_golden.py:make_rng (seeded keep-masks)."""
    g = torch.Generator().manual_seed(seed)

    def bern(shape, keep):
        return (torch.rand(shape, generator=g) < keep).float()

    rng = {"dec_keep": bern((n, 160, 16, 16), 0.8),
           "emb_keep": bern((n, 272, 8, 8), 1.0 - drop_rate),
           "feat_keep": bern((n, 1792), 1.0 - drop_rate),
           "drop_connect": {}}
    for idx in range(1, nblk):
        rng["drop_connect"][idx] = bern((n,), 1.0 - dc_rate * idx / nblk)
    return rng


def oracle_state(sf_coef=0.0, fuse_coef=0.3, dtype=torch.float32, requires_grad=False):
    sd = param_fill.fill_state_dict(eb4.eb4_state_shapes(2), sf_coef, fuse_coef, dtype)
    if requires_grad:
        for k, v in sd.items():
            if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")) \
                    and k != "bottleneck.bias":
                v.requires_grad_(True)
    return sd


SMOOTH_LAMBDAS = dict(LAMBDAS, lambda_recons=0.0, lambda_freq=0.0)   # drops the two L1 (sign-gradient) terms


def oracle_train_pass1(sd, x, tgt, rng, drop_rate=0.5, lam=None):
    out = eb4.forward_eb4(sd, x, training=True, drop_rate=drop_rate, rng=rng)
    n_real = int((tgt == 0).sum())
    ls = losses.pass1_loss(out, tgt, n_real, len(tgt) - n_real, lam or LAMBDAS)
    ls["total_loss"].backward()
    return out, ls
