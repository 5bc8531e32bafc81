"""TEST INFRASTRUCTURE — seeded stand-ins for the global-RNG draws of the pass-2 perturbation
(model/unidefense.py:177-198 and engine/abstract_engine.py:287-289 of the reference), shared by the golden generator
(oracle/make_golden_step2.py, run against the reference) and the GPU test (tests/test_d_engine_gpu.py, run around the HIP
engine): both sides then take the same branch with the same permutation lists and mixing coefficients.

  torch.rand(1)            -> 0.75 for the style branch ('freq' / 'efdm'), 0.0 for the PERT_FUNCS branch ('downscale')
  torch.randint(lo, hi, (1,)) -> 0 ('freq', 'downscale': first entry) or 1 ('efdm')
  torch.randperm(n)        -> a permutation from a seeded generator (call counter in the seed)
  torch.rand((B,1,1[,1]))  -> the transfer's lmda from a seeded generator
Other shapes fall through to the real functions."""
import contextlib

import torch


@contextlib.contextmanager
def pinned_draws(pert):
    assert pert in ("downscale", "freq", "efdm")
    orig = dict(rand=torch.rand, randint=torch.randint, randperm=torch.randperm)
    calls = {"perm": 0, "lmda": 0}

    def fake_rand(*size, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            size = tuple(size[0])
        if size == (1,):
            return torch.full((1,), 0.0 if pert == "downscale" else 0.75)
        if len(size) in (3, 4) and all(s == 1 for s in size[1:]) and "device" not in kw and "generator" not in kw:
            calls["lmda"] += 1
            return orig["rand"](size, generator=torch.Generator().manual_seed(777 + calls["lmda"]))
        return orig["rand"](*size, **kw)

    def fake_randint(low, high=None, size=None, **kw):
        v = 1 if pert == "efdm" else 0
        return torch.full(tuple(size) if size is not None else (1,), v, dtype=torch.int64)

    def fake_randperm(n, **kw):
        if "generator" in kw or "device" in kw:
            return orig["randperm"](n, **kw)
        calls["perm"] += 1
        return orig["randperm"](n, generator=torch.Generator().manual_seed(555 + calls["perm"]))

    torch.rand, torch.randint, torch.randperm = fake_rand, fake_randint, fake_randperm
    try:
        yield calls
    finally:
        torch.rand, torch.randint, torch.randperm = orig["rand"], orig["randint"], orig["randperm"]
