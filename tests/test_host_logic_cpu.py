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


def test_pixel_major_dft_matrices_against_torch_fft():
    """kernels._dft_pixel_mats: the matrices of the any-side rfft2 / irfft2 done as two batched GEMMs on the pixel-major tensor
    (the 95 x 95 map of the 380 x 380 trunk, model/efficientnet/exp.py:55-62).  The two GEMMs of each direction are restated here
    with einsum in float64 — ud_gemm's a_mode 1 reads A k-major: product = A[:, :M]^T B — and held to torch.fft with ud_rfft2's /
    ud_irfft2's contract (scale, column weight off the self-conjugate columns, Hermitian multiplicity).  Sides: 95 (odd), 14 (even:
    a Nyquist column), 13 (odd half width: the matrix rows are padded to a multiple of 4 and the slack must be zero)."""
    import torch
    from unidefense_amd import kernels as K
    g = torch.Generator().manual_seed(0)
    for S in (95, 14, 13):
        N, C = 2, 8
        Wh = S // 2 + 1
        scale, wi = 0.37, 1.7
        colw = torch.full((Wh,), wi, dtype=torch.float64)
        colw[0] = 1.0
        if S % 2 == 0:
            colw[-1] = 1.0
        x = torch.randn(N, S, S, C, generator=g, dtype=torch.float64)
        a1, a2 = [t.double() for t in K._dft_pixel_mats(S, scale, wi, False, "cpu")]
        assert a1.shape == (S, -(-2 * S // 4) * 4) and a2.shape == (2 * S, -(-2 * Wh // 4) * 4)
        assert float(a1[:, 2 * S:].abs().sum()) == 0.0 and float(a2[:, 2 * Wh:].abs().sum()) == 0.0
        T = torch.einsum("hm,nhq->nmq", a1[:, :2 * S], x.reshape(N, S, S * C)).reshape(N, S, 2 * S, C)      # [n][ky][(ri, w)][c]
        Y = torch.einsum("km,nykc->nymc", a2[:, :2 * Wh], T).reshape(N, S, Wh, 2 * C)
        ref = torch.fft.rfft2(x.permute(0, 3, 1, 2)) * (scale * colw)
        ref = torch.stack([ref.real, ref.imag], -1).permute(0, 2, 3, 4, 1).reshape(N, S, Wh, 2 * C)
        assert float((Y - ref).abs().max()) < 2e-5 * float(ref.abs().max())                                # fp32 matrices
        Yin = torch.randn(N, S, Wh, 2 * C, generator=g, dtype=torch.float64)
        a3, a4 = [t.double() for t in K._dft_pixel_mats(S, scale, wi, True, "cpu")]
        assert float(a3[:, 2 * S:].abs().sum()) == 0.0 and float(a4[:, S:].abs().sum()) == 0.0
        U = torch.einsum("km,nykc->nymc", a3[:, :2 * S], Yin.reshape(N, S, 2 * Wh, C))                      # [n][ky][(ri, w)][c]
        xo = torch.einsum("kh,nkq->nhq", a4[:, :S], U.reshape(N, 2 * S, S * C)).reshape(N, S, S, C)
        Yc = torch.complex(Yin.reshape(N, S, Wh, 2, C)[:, :, :, 0], Yin.reshape(N, S, Wh, 2, C)[:, :, :, 1]).permute(0, 3, 1, 2)
        refx = (torch.fft.irfft2(Yc * colw, s=(S, S), norm="forward") * scale).permute(0, 2, 3, 1)
        assert float((xo - refx).abs().max()) < 2e-5 * float(refx.abs().max())


def test_pair_and_mixed_precision_plan_rules():
    """host-side rules of the paired planes-GEMM launches (kernels.spectral_bwd) and of the mixed-precision mode's plans: which
    plans pair, which problem leads the grid, when a half-stored 1x1 conv goes to the planes kernel (only under ud_gemm path 3)."""
    from unidefense_amd import kernels as K, lib
    assert K._p3_pair_ok(("plain",), ("plain",)) and K._p3_pair_ok(("split", 3), ("split", 2)) and K._p3_pair_ok(("sk",), ("plain",))
    assert not K._p3_pair_ok(("tail", 1024, 2), ("plain",)) and not K._p3_pair_ok(("plain",), ("sk",))
    # a weight gradient with few long tiles leads: the 16 x 16 spectral conv (225 tiles, reduction 4608 > 2 x 1920) and the 32 x 32
    # one (36 tiles x split 4, 4352 rows per slice); not the 8 x 8 one (676 short tiles) nor a thin project conv
    assert K._p3_pair_tn_first(4608, 1920, 1920, ("plain",), ("plain",))
    assert K._p3_pair_tn_first(17408, 672, 672, ("plain",), ("split", 4))
    assert not K._p3_pair_tn_first(1280, 3264, 3264, ("plain",), ("plain",))
    assert not K._p3_pair_tn_first(2048, 1632, 272, ("plain",), ("split", 3))
    prev = lib.call("ud_gemm_get_path")
    try:
        lib.call("ud_gemm_set_path", 0)
        assert K._p1_plans_for(9216, 1920, 1920) is None                      # the fp32 mode never takes the one-plane form
        lib.call("ud_gemm_set_path", 3)
        pl = K._p1_plans_for(9216, 1920, 1920)
        assert pl is not None and pl["nt"] == ("plain",) and pl["nn"] == ("plain",)      # half results: no atomics, plain launches
        assert pl["tn"][0] in ("plain", "split")
        assert K._p1_plans_for(512, 1920, 1920) is None and K._p1_plans_for(9216, 64, 1920) is None      # below the thresholds
        assert K._p1_plans_for(9216, 1920, 1924) is None                      # K % 8: whole 16-byte groups of half values
    finally:
        lib.call("ud_gemm_set_path", prev)
