"""CPU: host-side launch planning (no kernel is launched) and index rules shared with the oracle."""
import numpy as np


def test_tail_plan_covers_rows_exactly_and_only_when_it_pays():
    from unidefense_amd import kernels as K
    assert K._tail_plan(4608, 1920, 1920) == (4352, 7)         # 36 x 15 = 540 tiles: 34 row-tiles plain + 2 split
    assert K._tail_plan(1280, 3264, 3264) == (1152, 8)         # 260 tiles: 9 row-tiles plain + 1 split 8
    assert K._tail_plan(17408, 672, 672) == (16384, 2)
    assert K._tail_plan(4096, 4096, 4096) is None              # 1024 tiles: whole rounds
    assert K._tail_plan(4608, 1344, 1344) is None              # 396 tiles: the tail round is more than a quarter full
    assert K._tail_plan(512, 1920, 1920) is None and K._tail_plan(4608, 64, 1920) is None
    for M in range(1024, 9000, 384):
        for N in (256, 1344, 1920, 3264):
            plan = K._tail_plan(M, N, 2048)
            if plan is None:
                continue
            m1, split = plan
            nt = -(-N // 128)
            assert 0 < m1 < M and m1 % 128 == 0 and 2 <= split <= 8
            assert (m1 // 128) * nt <= (-(-M // 128) * nt // 256) * 256          # the plain part fits whole rounds
            assert (-(-(M - m1) // 128)) * nt * split <= 256                      # the split part fits one round


def test_split_rules_are_sane():
    from unidefense_amd import kernels as K
    assert K._pick_split(600, 100000) == 1 and K._pick_split(2, 256) == 1
    s = K._pick_split(2, 524288)
    assert 2 <= s <= 1024 and 524288 // s >= 256
    assert K._fwd_split(131072, 192, 32) == 1                  # plenty of tiles
    assert K._fwd_split(2048, 272, 2448) >= 2                  # 8x8 stage: under-filled launch


def test_downscale_index_matches_the_oracle_rule():
    from oracle import perturb as OP
    from unidefense_amd.model import perturb as P
    for s in (64, 128, 256, 320):
        assert np.array_equal(P.downscale_index(s), OP.downscale_index(s))


def test_gemm_tuner_candidates_and_plan_cache(tmp_path):
    """The per-shape GEMM tuner's host logic: candidate (tile, split-K) grids respect the kernel's limits (reduction rows per
    split, workgroup count), the heavy spectral shapes get the short list, and the plan cache file round-trips."""
    import importlib
    import json
    import os
    from unidefense_amd import kernels as K
    for M, N, Kd in ((2048, 272, 1632), (131072, 32, 192), (1152, 3264, 3264), (160, 960, 8192), (32, 144, 131072)):
        cands = K._tune_candidates(M, N, Kd)
        assert cands and len(set(cands)) == len(cands)
        heavy = 2.0 * M * N * Kd > 2.0e10
        for cfg, split in cands:
            bm, bn = K._X3_TILES[cfg]
            tiles = -(-M // bm) * -(-N // bn)
            assert 1 <= cfg <= 4 and split >= 1
            if split > 1:
                assert Kd // split >= 64 and tiles * split <= 4096
            if heavy:
                assert split <= 4 and cfg != 4
        assert any(s == 1 for _, s in cands) or Kd >= 1024
    # cache file: written on every new plan, read back at import
    path = str(tmp_path / "plans.json")
    from unidefense_amd.config import cfg
    old = cfg.gemm_tune_cache
    cfg.gemm_tune_cache = path
    try:
        K2 = importlib.reload(K)
        K2._TUNED[("nt", 2048, 272, 1632, False, 0)] = (4, 3)
        K2._TUNED[("tn", 160, 960, 8192, False, 0)] = None
        K2._tune_cache_save()
        assert json.load(open(path))
        K3 = importlib.reload(K2)
        assert K3._TUNED[("nt", 2048, 272, 1632, False, 0)] == (4, 3) and K3._TUNED[("tn", 160, 960, 8192, False, 0)] is None
    finally:
        cfg.gemm_tune_cache = old
        importlib.reload(K)


def test_tape_cast_converts_values_and_gradients():
    """tape.cast — the storage-type boundary of the half-storage trunk: y = x.to(dtype), the gradient comes back in x's type
    and accumulates with the other consumers' gradients of x."""
    import torch
    from unidefense_amd import tape as T
    tape = T.Tape()
    x = torch.randn(2, 3, 3, 4)
    y = T.cast(tape, x, torch.float16)
    assert y.dtype == torch.float16 and torch.equal(y, x.half())
    assert T.cast(tape, x, torch.float32) is x                       # same type: identity, no tape node
    n_nodes = len(tape.nodes)
    assert n_nodes == 1
    gy = torch.randn(2, 3, 3, 4).half()
    tape.add_grad(y, gy)
    for fn in reversed(tape.nodes):
        fn()
    gx = tape.pop_grad(x)
    assert gx.dtype == torch.float32 and torch.equal(gx, gy.float())
