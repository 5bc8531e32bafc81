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
