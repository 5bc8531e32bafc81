"""TEST INFRASTRUCTURE — deterministic, key-name-seeded parameter filler.

The same state-dict can be regenerated anywhere (this container with the reference
imported, or the GPU box without it) from nothing but the key names and shapes, so
golden vectors only have to carry inputs/outputs, never the 513 MB of weights.
(SURVEY.md §8c "Parameter filler".)
"""
import zlib
import math
import torch


def _gen(key: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(key.encode("utf-8")) & 0x7FFFFFFF)
    return g


def fill_value(key: str, shape, sf_coef: float = -10.0, fuse_coef: float = 0.0,
               dtype=torch.float32) -> torch.Tensor:
    """Value for one state-dict entry, as a CPU tensor of ``dtype``.

    Rules (by key suffix / rank):
      * ``num_batches_tracked``                -> 0 (int64)
      * ``sf_coef`` / ``fuse_coef`` (0-dim)    -> the given constants
      * ``running_mean``                       -> 0.1 z
      * ``running_var``                        -> 1 + 0.1 |z|
      * norm ``weight`` (1-D)                  -> 1 + 0.1 z
      * any ``bias`` (1-D)                     -> 0.1 z
      * conv / linear weight (>= 2-D)          -> z * sqrt(2 / fan_in)
    All z are N(0,1) draws from a generator seeded with crc32(key), generated in
    float32 (so float64 callers see exactly the float32 values).
    """
    shape = tuple(shape)
    if key.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.int64)
    if key.endswith("sf_coef"):
        return torch.full(shape, sf_coef, dtype=dtype)
    if key.endswith("fuse_coef"):
        return torch.full(shape, fuse_coef, dtype=dtype)
    z = torch.randn(shape, generator=_gen(key), dtype=torch.float32)
    if key.endswith("running_mean"):
        v = 0.1 * z
    elif key.endswith("running_var"):
        v = 1.0 + 0.1 * z.abs()
    elif len(shape) <= 1:
        if key.endswith("bias"):
            v = 0.1 * z
        else:
            v = 1.0 + 0.1 * z
    else:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        v = z * math.sqrt(2.0 / fan_in)
    return v.to(dtype)


def fill_state_dict(shapes: dict, sf_coef: float = -10.0, fuse_coef: float = 0.0,
                    dtype=torch.float32) -> dict:
    """shapes: {key: shape}.  Returns {key: tensor}."""
    return {k: fill_value(k, s, sf_coef, fuse_coef, dtype) for k, s in shapes.items()}


def fill_module_(module: torch.nn.Module, sf_coef: float = -10.0, fuse_coef: float = 0.0):
    """In-place fill of every parameter and buffer of ``module`` (any device)."""
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        t = fill_value(k, v.shape, sf_coef, fuse_coef,
                       dtype=v.dtype if v.dtype.is_floating_point else torch.float32)
        new[k] = t.to(device=v.device, dtype=v.dtype)
    module.load_state_dict(new, strict=True)
    return module


def make_input(n: int, size: int, seed: int = 0, dtype=torch.float32) -> torch.Tensor:
    """x = 2*U(0,1) - 1, shape [n,3,size,size]  (SURVEY.md §8d synthetic inputs)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return (2.0 * torch.rand(n, 3, size, size, generator=g, dtype=torch.float32) - 1.0).to(dtype)


def make_labels(n: int) -> torch.Tensor:
    """[0]*n/2 + [1]*n/2 — real samples first (loss/triplet_loss.py:48-53)."""
    return torch.tensor([0] * (n // 2) + [1] * (n - n // 2), dtype=torch.int64)
