"""GPU parity tests of ud_gemm_p3 (csrc/gemm_p3.hip) and its producers, through the C ABI: the GEMM on pre-split operands that
serves the spectral 1x1 convs (model/efficientnet/exp.py:57: F.conv2d(x_freq, freq_conv.weight), its data gradient and its
weight gradient).

* prec 3 (three bf16 planes): the arithmetic of ud_gemm's split-bf16 kernel with the split done by the producer — results must
  be BITWISE those of ud_gemm (plain launches; split-K launches are compared with float64).
* prec 2 (two fp16 planes, power-of-two scale per tensor or per row): against float64, error relative to sum |a||b| of every
  output — the bar is ud_gemm's own error on the same operands (x 2) or 2e-6, on well- and badly-conditioned operands.
* ud_split_planes* / ud_absmax: the planes re-assembled on the host must give back the input (prec 3: exactly; prec 2: to
  2^-21 of the scale's maximum), the scale must put the maximum into [2^14, 2^15).
* stream-K, split-K (atomics and ordered slices), tail plan, row offsets, epilogue statistics.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _operands(kind, M, N, Kd, dev, dist="randn"):
    """(a, b, a_mode, b_mode) in the storage of the three products.  k-e4 / rows-e3: magnitudes e^(4 randn) along k, e^(3 randn)
    across the GEMM rows (clamped to 2^+-8.7 around the median: inside the 2^18 full-precision window of ONE tensor scale); tiny / huge scale A only."""
    def mk(r, c, k_dim, first):
        g = torch.randn(r, c, device=dev)
        if dist == "same-sign":
            return g.abs()
        if dist == "lognormal3":
            return g * torch.exp(3 * torch.randn(r, c, device=dev))
        if dist in ("k-e4", "rows-e3"):
            along_k = dist == "k-e4"
            sig = 4.0 if along_k else 3.0
            shape = ((1, c) if k_dim == 1 else (r, 1)) if along_k else ((r, 1) if k_dim == 1 else (1, c))
            return g * torch.exp((sig * torch.randn(shape, device=dev)).clamp(-2 * sig, 2 * sig))
        if dist == "tiny" and first:
            return g * 1e-30
        if dist == "huge" and first:
            return g * 1e30
        if dist == "zero-rows":
            g[::3] = 0
        return g
    if kind == "nt":
        return mk(M, Kd, 1, True), mk(N, Kd, 1, False), 0, 0
    if kind == "nn":
        return mk(M, Kd, 1, True), mk(Kd, N, 0, False), 0, 1
    return mk(Kd, M, 0, True), mk(Kd, N, 0, False), 1, 1


def _ref64(kind, a, b):
    a, b = a.double(), b.double()
    return a @ b.t() if kind == "nt" else a @ b if kind == "nn" else a.t() @ b


def _x3(K, kind, a, b, M, N, Kd):
    am, bm = {"nt": (0, 0), "nn": (0, 1), "tn": (1, 1)}[kind]
    out = torch.empty(M, N, device=a.device)
    return K._gemm(a, b, out, M, N, Kd, a.shape[1], b.shape[1], N, am, bm, 0, 1, cfg=1)


@pytest.mark.parametrize("kind,M,N,Kd", [("nt", 256, 256, 64), ("nt", 384, 200, 96), ("nn", 300, 264, 160),
                                         ("tn", 192, 320, 256), ("nt", 1280, 3264, 3264 // 3), ("tn", 672, 672, 4352)])
def test_prec3_is_bitwise_the_in_kernel_split(kind, M, N, Kd):
    from unidefense_amd import kernels as K
    dev = _dev()
    torch.manual_seed(1)
    a, b, am, bm = _operands(kind, M, N, Kd, dev)
    ap, bp = K.split_planes(a), K.split_planes(b)
    out = torch.full((M, N), float("nan"), device=dev)
    K._gemm_p3(ap, bp, out, M, N, Kd, am, bm)
    ref = _x3(K, kind, a, b, M, N, Kd)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    # split-K (ordered slices under cfg.deterministic, else atomics) and stream-K against float64
    want = _ref64(kind, a, b)
    scale = _ref64(kind, a.abs(), b.abs())
    for out_mode, split, cfg in ((2, 3, 0), (0, 1, 0x800)):
        if Kd // 32 < split:
            continue
        o = K.split_out((M, N), a) if out_mode == 2 else torch.zeros(M, N, device=dev)
        K._gemm_p3(ap, bp, o, M, N, Kd, am, bm, out_mode, split, cfg=cfg)
        torch.cuda.synchronize()
        assert ((o.double() - want).abs() / scale).max().item() < 2e-6


@pytest.mark.parametrize("dist", ["randn", "same-sign", "lognormal3", "k-e4", "rows-e3", "tiny", "huge", "zero-rows"])
@pytest.mark.parametrize("kind,M,N,Kd", [("nt", 384, 200, 96), ("nn", 640, 1344, 1344), ("tn", 672, 416, 2176)])
def test_prec2_has_fp32_gemm_accuracy(dist, kind, M, N, Kd):
    from unidefense_amd import kernels as K
    from tests.margins import within
    dev = _dev()
    torch.manual_seed(2)
    a, b, am, bm = _operands(kind, M, N, Kd, dev, dist)
    want = _ref64(kind, a, b)
    scale = _ref64(kind, a.abs(), b.abs()) + 1e-300
    e3 = ((_x3(K, kind, a, b, M, N, Kd).double() - want).abs() / scale).max().item()
    ap, bp = K.split_planes(a, prec=2), K.split_planes(b, prec=2)
    worst = 0.0
    for out_mode, split, cfg in ((0, 1, 0), (2, 2, 0), (0, 1, 0x800)):
        o = K.split_out((M, N), a) if out_mode == 2 else torch.zeros(M, N, device=dev)
        K._gemm_p3(ap, bp, o, M, N, Kd, am, bm, out_mode, split, cfg=cfg)
        torch.cuda.synchronize()
        assert torch.isfinite(o).all()
        worst = max(worst, ((o.double() - want).abs() / scale).max().item())
    assert within(f"gemm_p3 prec 2 {kind} {dist} vs float64 (bar: the in-kernel-split GEMM's error x 2)", worst, max(2e-6, 2 * e3))


def test_prec2_row_scales_cover_rows_of_any_magnitude():
    """one scale per GEMM row (mode 0): rows 2^+-35 apart keep fp32-GEMM accuracy each (a single tensor scale could not)"""
    from unidefense_amd import kernels as K
    dev = _dev()
    torch.manual_seed(3)
    M, N, Kd = 512, 384, 672
    a = torch.randn(M, Kd, device=dev) * torch.exp(8 * torch.randn(M, 1, device=dev))
    b = torch.randn(N, Kd, device=dev) * torch.exp(8 * torch.randn(N, 1, device=dev))
    ap, bp = K.split_planes(a, prec=2, per_row=True), K.split_planes(b, prec=2, per_row=True)
    out = torch.empty(M, N, device=dev)
    K._gemm_p3(ap, bp, out, M, N, Kd, 0, 0)
    torch.cuda.synchronize()
    want = a.double() @ b.double().t()
    scale = a.double().abs() @ b.double().abs().t()
    assert ((out.double() - want).abs() / scale).max().item() < 2e-6


@pytest.mark.parametrize("R,C", [(300, 96), (1280, 3264), (64, 40)])
def test_split_kernels_reassemble_to_the_input(R, C):
    from unidefense_amd import kernels as K
    dev = _dev()
    torch.manual_seed(4)
    x = torch.randn(R, C, device=dev) * torch.exp(2 * torch.randn(R, C, device=dev))

    def unpack(pl, dtype):
        npan, rows = pl.npanel, pl.panel // 32
        planes = pl.buf.view(pl.prec, npan, rows, 32)[:, :, :R].view(dtype)          # [prec][panel][R][32]
        return planes.permute(0, 2, 1, 3).reshape(pl.prec, R, npan * 32)[:, :, :C].double()
    p3 = K.split_planes(x)
    torch.cuda.synchronize()
    assert torch.equal(unpack(p3, torch.bfloat16).sum(0), x.double())          # exact three-way split
    for per_row in (False, True):
        p2 = K.split_planes(x, prec=2, per_row=per_row)
        torch.cuda.synchronize()
        h = unpack(p2, torch.float16)
        inv = p2.inv.double().view(-1, 1) if per_row else p2.inv.double()
        back = (h[0] + h[1] / 2048.0) * inv
        top = x.abs().amax(1, keepdim=True).double() if per_row else x.abs().max().double()
        assert ((back - x.double()).abs() / top).max().item() < 2.0 ** -21
        smax = (x.double().abs() / inv).amax(1) if per_row else (x.double().abs() / inv).max()
        assert (smax >= 2.0 ** 14).all() and (smax < 2.0 ** 15).all()
        if (C % 32) != 0:          # the last panel's padding columns are zeros (they are multiplied as K padding)
            pad = p2.buf.view(2, p2.npanel, p2.panel // 32, 32)[:, -1, :R, C % 32:]
            assert (pad == 0).all()


def test_row_offsets_tail_plan_and_epilogue_statistics():
    from unidefense_amd import kernels as K
    dev = _dev()
    torch.manual_seed(5)
    M, N, Kd = 640, 256, 128
    a, b = torch.randn(M, Kd, device=dev), torch.randn(N, Kd, device=dev)
    for prec in (3, 2):
        ap, bp = K.split_planes(a, prec=prec), K.split_planes(b, prec=prec)
        out = torch.empty(256, N, device=dev)
        K._gemm_p3(ap, bp, out, 256, N, Kd, 0, 0, a_row0=384)
        acc = torch.zeros(2 * N, dtype=torch.float64, device=dev)
        full = torch.empty(M, N, device=dev)
        _, done = K._gemm_p3(ap, bp, full, M, N, Kd, 0, 0, stats=acc)
        torch.cuda.synchronize()
        want = a.double() @ b.double().t()
        assert (out.double() - want[384:]).abs().max().item() < 1e-4
        assert done
        assert ((acc[:N] - full.double().sum(0)).abs().max() / full.double().abs().sum(0).max()).item() < 1e-12
        assert ((acc[N:] - (full.double() ** 2).sum(0)).abs().max() / (full.double() ** 2).sum(0).max()).item() < 1e-12
    # the three-product context the tape uses, every plan kind, against the in-kernel-split functions
    from unidefense_amd.config import override
    M, C2 = 1152, 672
    x, w, dy = torch.randn(M, C2, device=dev), torch.randn(C2, C2, device=dev) * 0.05, torch.randn(M, C2, device=dev)
    with override(spectral_p2="off"):
        y0, c0 = K.spectral_fwd(x, w)
        dx0, dw0 = K.spectral_dgrad(c0, dy), K.spectral_wgrad(c0, dy)
    with override(spectral_p2="on"):
        y1, c1 = K.spectral_fwd(x, w)
        assert c1.plans is not None
        dx1, dw1 = K.spectral_dgrad(c1, dy), K.spectral_wgrad(c1, dy)
    torch.cuda.synchronize()
    for got, ref in ((y1, y0), (dx1, dx0), (dw1, dw0)):
        assert ((got - ref).abs().max() / ref.abs().max()).item() < 5e-6
    xp, wp = K.split_planes(x, prec=2), K.split_planes(w, prec=2)
    for kind, (m, n, k, pa, pb, ref) in {"nt": (M, C2, C2, xp, wp, y0), "nn": (M, C2, C2, xp, wp, None),
                                         "tn": (C2, C2, M, xp, xp, None)}.items():
        if ref is None:
            ref = (x.double() @ w.double()).float() if kind == "nn" else (x.double().t() @ x.double()).float()
        for plan in K._p2_plans(kind, m, n, k) + [("tail", 1024, 2)] * (kind != "tn"):
            got = K._p2_run(kind, plan, pa, pb, m, n, k, x)
            torch.cuda.synchronize()
            assert ((got - ref).abs().max() / ref.abs().max()).item() < 5e-6, (kind, plan)


@pytest.mark.parametrize("kind,M,N,Kd", [("nt", 128 * 11 + 40, 128 * 5, 96), ("tn", 128 * 7, 128 * 9 + 8, 512), ("nn", 128 * 3, 128 * 2, 64)])
def test_xcd_raster_is_a_permutation_of_the_tiles(kind, M, N, Kd):
    """tile_cfg bit 9 only changes WHICH workgroup computes which tile (groups of GM tile rows, one contiguous range per XCD):
    every tile still computed exactly once — the result is bitwise the round-robin deal's, for tile counts that are not
    multiples of 8 or of GM, plain and split-K (ordered slices), statistics included."""
    from unidefense_amd import kernels as K
    dev = _dev()
    torch.manual_seed(6)
    a, b, am, bm = _operands(kind, M, N, Kd, dev)
    ap, bp = K.split_planes(a, prec=2), K.split_planes(b, prec=2)
    saved = K._P3_RASTER
    try:
        outs = []
        for ras in (0,) + tuple(0x200 | gm << 12 for gm in (1, 2, 3, 4, 7, 8, 15)):
            K._P3_RASTER = ras
            o = torch.full((M, N), float("nan"), device=dev)
            acc = torch.zeros(2 * N, dtype=torch.float64, device=dev)
            K._gemm_p3(ap, bp, o, M, N, Kd, am, bm, stats=acc if kind == "nt" else None)
            s = K.split_out((M, N), a)
            K._gemm_p3(ap, bp, s, M, N, Kd, am, bm, 2, 2)
            torch.cuda.synchronize()
            outs.append((o, s.clone(), acc))
        for o, s, acc in outs[1:]:
            assert torch.equal(o, outs[0][0]) and (s - outs[0][1]).abs().max().item() <= 1e-5 * outs[0][1].abs().max().item()
            assert ((acc - outs[0][2]).abs().max() <= 1e-12 * outs[0][2].abs().max().clamp_min(1e-300)).item()
    finally:
        K._P3_RASTER = saved


@pytest.mark.parametrize("M,N,Kd,pn,pt,acc", [(128 * 7 + 40, 128 * 3, 128 * 5, ("plain",), ("plain",), False),
                                              (128 * 9, 128 * 2 + 16, 128 * 4 + 32, ("plain",), ("split", 3), True),
                                              (128 * 5, 128 * 6, 128 * 3, ("split", 2), ("split", 2), False),
                                              (4608, 1920, 1920, ("plain",), ("plain",), False),
                                              (1280, 3264, 3264, ("sk",), ("plain",), False)])
def test_data_and_weight_gradient_in_one_launch(M, N, Kd, pn, pt, acc):
    """ud_gemm_p3_pair (round 5): the data gradient dx[M, K] = dy . w and the weight gradient dw[N, K] = dy^T . x of a 1x1 conv as
    ONE grid — the weight gradient's workgroups follow the data gradient's.  Every tile is computed by the same code on the same
    operands as in two ud_gemm_p3 launches: plain stores are bitwise equal, atomic split-K sums agree to the order of the adds;
    ragged tile edges, an existing term to add onto, both problems with their own XCD raster."""
    from unidefense_amd import kernels as K
    dev = _dev()
    torch.manual_seed(11)
    x, w, dy = torch.randn(M, Kd, device=dev), torch.randn(N, Kd, device=dev) * 0.05, torch.randn(M, N, device=dev)
    term = torch.randn(M, Kd, device=dev) if acc else None
    ctx = K.SpectralCtx()
    ctx.M, ctx.N, ctx.K, ctx.dy = M, N, Kd, None
    ctx.x, ctx.w = K.split_planes(x, prec=2), K.split_planes(w, prec=2)
    ctx.plans = {"nt": ("plain",), "nn": pn, "tn": pt}
    saved = K._P3_PAIR
    try:
        K._P3_PAIR = False
        dx0, dw0 = K.spectral_bwd(ctx, dy, out=term.clone() if acc else None)
        K._P3_PAIR = True
        dx1, dw1 = K.spectral_bwd(ctx, dy, out=term.clone() if acc else None)
        torch.cuda.synchronize()
    finally:
        K._P3_PAIR = saved
    for got, ref, plan in ((dx1, dx0, pn), (dw1, dw0, pt)):
        if plan[0] == "plain":          # (a stream-K reference sums its K segments by atomics)
            assert torch.equal(got, ref)
        else:
            assert ((got - ref).abs().max() / ref.abs().max()).item() < 1e-5
    want = dy.double() @ w.double() + (term.double() if acc else 0.0)
    assert ((dx1.double() - want).abs().max() / want.abs().max()).item() < 5e-6
    want = dy.double().t() @ x.double()
    assert ((dw1.double() - want).abs().max() / want.abs().max()).item() < 5e-6
    assert K._p3_pair_ok(("plain",), ("split", 2)) and K._p3_pair_ok(("sk",), ("plain",))      # (stream-K: as plain tiles in the pair)
    assert not K._p3_pair_ok(("tail", 1024, 2), ("plain",)) and not K._p3_pair_ok(("plain",), ("sk",))


@pytest.mark.parametrize("kind,M,N,Kd", [("nt", 128 * 9 + 32, 128 * 5, 512), ("nn", 128 * 6, 128 * 3 + 64, 768), ("tn", 128 * 4, 128 * 5, 2048)])
def test_prec1_is_the_fp16_operand_gemm(kind, M, N, Kd):
    """ud_gemm_p3 prec 1 (round 5; the mixed-precision mode, BASELINE configs[4]): ONE fp16 plane per operand, one product per tile,
    fp32 accumulation.  Its operands: a half-stored activation through ud_planes_from_half (a layout pass, values unchanged) and
    the FIRST plane of prec-2 planes (weights).  Held to the float64 product of the fp16-ROUNDED operands to fp32-accumulation
    accuracy (1e-5 of the result's scale), fp32 and half-stored results, ragged edges; the layout pass round-trips bit for bit."""
    from unidefense_amd import kernels as K
    dev = _dev()
    torch.manual_seed(7)
    a, b, am, bm = _operands(kind, M, N, Kd, dev)
    ah = a.half()
    ap = K.planes_from_half(ah)                                    # activation operand: one plane, scale 1
    bp = K.split_planes(b.contiguous(), prec=2)                    # weight-like operand: prec-2 planes, first plane read
    R, Cc = ah.shape
    plane0 = ap.buf.view(ap.npanel, ap.panel // 32, 32)[:, :R].view(torch.float16).permute(1, 0, 2).reshape(R, ap.npanel * 32)
    torch.cuda.synchronize()
    assert torch.equal(plane0[:, :Cc], ah) and float(ap.inv) == 1.0
    binv = float(bp.inv)
    b0 = bp.buf.view(2, bp.npanel, bp.panel // 32, 32)[0, :, :b.shape[0]].view(torch.float16).permute(1, 0, 2).reshape(b.shape[0], bp.npanel * 32)
    b_eff = b0[:, :b.shape[1]].double() * binv                     # what the kernel multiplies by
    if kind == "nt":
        want = ah.double() @ b_eff.t()
    elif kind == "nn":
        want = ah.double() @ b_eff
    else:
        want = ah.double().t() @ b_eff
    o32 = torch.full((M, N), float("nan"), device=dev)
    K._gemm_p3(ap, bp, o32, M, N, -(-Kd // 32) * 32, am, bm)
    torch.cuda.synchronize()
    assert ((o32.double() - want).abs().max() / want.abs().max()).item() < 1e-5
    if kind != "tn":
        o16 = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
        K._gemm_p3(ap, bp, o16, M, N, -(-Kd // 32) * 32, am, bm)
        term = torch.randn(M, N, device=dev).half()
        acc = term.clone()
        K._gemm_p3(ap, bp, acc, M, N, -(-Kd // 32) * 32, am, bm, 1)
        torch.cuda.synchronize()
        assert torch.equal(o16, o32.half())
        assert torch.equal(acc, (o32 + term.float()).half())


def test_weight_planes_of_a_step_in_two_launches():
    """ud_split_planes_h2t_multi (kernels._WeightPlaneBatch): the planes of all registered weight matrices from one absmax and
    one split launch equal the per-matrix ud_absmax + ud_split_planes_h2t planes BITWISE (scale included) for assorted shapes
    (rows not a multiple of 64, columns not a multiple of 32, one-block and 256-block matrices); they are handed out only while
    the parameter is unchanged — after an in-place update the caller gets a fresh split, the next begin_forward() re-splits."""
    from unidefense_amd import kernels as K
    dev = _dev()
    torch.manual_seed(7)
    batch = K._WeightPlaneBatch()
    saved = K._WEIGHT_PLANES
    K._WEIGHT_PLANES = batch
    try:
        shapes = [(1920, 1920), (160, 960), (272, 1632), (672, 112), (64, 64), (3264, 3264), (100, 36), (36, 100)]
        params = [torch.nn.Parameter(torch.randn(r, c, device=dev) * (10.0 ** (i - 3))) for i, (r, c) in enumerate(shapes)]
        views = [p.view(p.shape[0], p.shape[1]) for p in params]

        def rows(pl):
            return pl.buf.view(2, pl.npanel, pl.panel // 32, 32)[:, :, :pl.R]
        for v in views:
            assert batch.lookup(v) is None
            K.weight_planes(v)                                  # first use: its own split, and it registers
        assert len(batch.entries) == len(shapes)
        K.begin_forward()
        torch.cuda.synchronize()
        for v in views:
            got, ref = batch.lookup(v), K.split_planes(v, prec=2)
            torch.cuda.synchronize()
            assert got is not None and torch.equal(rows(got), rows(ref)) and torch.equal(got.inv, ref.inv), tuple(v.shape)
            assert K.weight_planes(v) is got
        K.end_forward()                                         # outside the forward that made them: not handed out
        assert all(batch.lookup(v) is None for v in views)
        K.begin_forward()
        # an optimizer-like in-place update: the batch's planes are stale and must not be handed out
        with torch.no_grad():
            params[1].mul_(3.0)
        assert batch.lookup(views[1]) is None and batch.lookup(views[0]) is not None
        fresh = K.weight_planes(views[1])
        K.begin_forward()
        torch.cuda.synchronize()
        again = batch.lookup(views[1])
        assert again is not None and torch.equal(rows(again), rows(fresh)) and torch.equal(again.inv, fresh.inv)
        # a parameter that goes away leaves the table
        del params[2:4], views[2:4], v, got, ref
        import gc
        gc.collect()
        K.begin_forward()
        torch.cuda.synchronize()
        assert len(batch.entries) == len(shapes) - 2
        for v in views:
            got, ref = batch.lookup(v), K.split_planes(v, prec=2)
            torch.cuda.synchronize()
            assert got is not None and torch.equal(rows(got), rows(ref))
    finally:
        K._WEIGHT_PLANES = saved


def test_weight_plane_batch_survives_dead_and_moved_parameters():
    """every registered parameter gone -> begin_forward() is a no-op (no empty table); a parameter whose storage was replaced
    (p.data = ...) leaves the table and re-registers on its next use"""
    from unidefense_amd import kernels as K
    dev = _dev()
    batch = K._WeightPlaneBatch()
    saved = K._WEIGHT_PLANES
    K._WEIGHT_PLANES = batch
    try:
        p = torch.nn.Parameter(torch.randn(128, 64, device=dev))
        K.weight_planes(p.view(128, 64))
        K.begin_forward()
        assert batch.lookup(p.view(128, 64)) is not None
        p.data = torch.randn(128, 64, device=dev)                  # new storage, same version
        assert batch.lookup(p.view(128, 64)) is None
        K.begin_forward()
        assert len(batch.entries) == 0
        ref = K.split_planes(p.view(128, 64), prec=2)
        got = K.weight_planes(p.view(128, 64))                     # own split; registers again
        torch.cuda.synchronize()
        assert torch.equal(got.buf, ref.buf) and len(batch.entries) == 1
        del p, got, ref
        import gc
        gc.collect()
        K.begin_forward()
        K.begin_forward()
        assert len(batch.entries) == 0
    finally:
        K._WEIGHT_PLANES = saved


@pytest.mark.parametrize("N,S,C", [(2, 8, 64), (3, 16, 48), (2, 32, 16), (32, 8, 1632), (2, 12, 32)])
def test_rfft2_writes_the_gemm_planes_itself(N, S, C):
    """csrc/fft.hip PlanesOut (round 5): ud_rfft2_ex_planes writes rfft2(act(bn(x))) straight into the fp16 x 2 planes of the
    spectral GEMM, scaled by a power of two taken from an a-priori BOUND of |Y| (count (gamma^2 + beta^2) per channel) instead of the
    exact maximum a pass over the result would give.  Checked: (1) the planes re-assemble to ud_rfft2_ex's fp32 result — every
    element to 2^-21 of the tensor's maximum times the bound's looseness, the LARGE elements (within 2^-10 of the maximum) to 22 bits
    of their own value; (2) nothing overflows: the scaled maximum stays below 2^15; (3) the bound is within 2^7 of the true maximum
    on these inputs (N <= 32: ~2 sqrt(N) expected); (4) the GEMM on these planes agrees with the GEMM on the split of the fp32
    result to 1e-6 of its scale; (5) the activated input written on the side is the one ud_rfft2_ex writes."""
    from unidefense_amd import kernels as K
    dev = _dev()
    K.reset_zero_pool()
    g = torch.Generator().manual_seed(S * 1000 + C + N)
    x = torch.randn(N, S, S, C, generator=g).to(dev)
    gamma = (1.0 + 0.3 * torch.randn(C, generator=g)).to(dev)
    beta = (0.2 * torch.randn(C, generator=g)).to(dev)
    M = N * S * S
    acc = K.zeros64(2 * C, x)
    K.colstats(x.view(M, C), acc)
    bn = K.DeferredBN(acc, C, M, gamma, beta, 1e-3, 1)
    sf = 1.0 / S
    Y, a_ref = K.rfft2_ex(x, sf, 1.0, bn=bn, want_act=True)
    pl, a = K.rfft2_ex_planes(x, sf, 1.0, bn=bn, want_act=True)
    torch.cuda.synchronize()
    assert torch.equal(a, a_ref)
    R = N * S * (S // 2 + 1)
    assert pl.R == R and pl.C == 2 * C
    h = pl.buf.view(2, pl.npanel, pl.panel // 32, 32)[:, :, :R].view(torch.float16).permute(0, 2, 1, 3).reshape(2, R, 2 * C).double()
    inv = float(pl.inv)
    back = (h[0] + h[1] / 2048.0) * inv
    Yd = Y.view(R, 2 * C).double()
    top = float(Yd.abs().max())
    scaled_max = top / inv
    loose = 2.0 ** 15 / scaled_max          # how far above the exact maximum the bound sits (>= 1)
    assert math.log2(1.0 / inv) == round(math.log2(1.0 / inv)) and scaled_max < 2.0 ** 15
    assert 1.0 <= loose < 2.0 ** 7, loose
    err = (back - Yd).abs()
    assert float(err.max()) <= 2.0 ** -21 * top * loose
    big = Yd.abs() >= top * 2.0 ** -10
    assert float((err[big] / Yd.abs()[big]).max()) < 2.0 ** -21
    # the product: planes from the transform vs planes split from the fp32 result, the same weights
    w = (torch.randn(2 * C, 2 * C, generator=g) / math.sqrt(2 * C)).to(dev)
    wp = K.split_planes(w, prec=2)
    ref = Yd @ w.double().t()
    got = K._gemm_p3(pl, wp, torch.empty(R, 2 * C, device=dev), R, 2 * C, -(-2 * C // 32) * 32, 0, 0)
    got2 = K._gemm_p3(K.split_planes(Y.view(R, 2 * C), prec=2), wp, torch.empty(R, 2 * C, device=dev), R, 2 * C, -(-2 * C // 32) * 32, 0, 0)
    scale = float(ref.abs().max())
    e1, e2 = float((got.double() - ref).abs().max()) / scale, float((got2.double() - ref).abs().max()) / scale
    print(f"  bound / max = 2^{math.log2(loose):.1f};  GEMM error / scale: planes from the transform {e1:.2e}, from the split {e2:.2e}")
    assert e1 < 1e-6 and e1 < 4 * e2 + 1e-7
    # ---- the backward's use: no BatchNorm in front, a per-channel ENERGY bound handed in (what ud_normbwd_apply_mix sums), the
    # gate factor sigmoid(alpha) applied to the result and to the bound, interior columns doubled
    dd = torch.randn(N, S, S, C, generator=g).to(dev) * torch.exp(torch.randn(C, generator=g)).to(dev)
    alpha = torch.tensor([-3.0], device=dev)
    en = (dd.double() ** 2).sum((0, 1, 2)) * 1.0001
    Y2, _ = K.rfft2_ex(dd, sf, 2.0, gate_alpha=alpha, gate_mode=1)
    pl2, _ = K.rfft2_ex_planes(dd, sf, 2.0, gate_alpha=alpha, gate_mode=1, energy=en.contiguous())
    torch.cuda.synchronize()
    h2 = pl2.buf.view(2, pl2.npanel, pl2.panel // 32, 32)[:, :, :R].view(torch.float16).permute(0, 2, 1, 3).reshape(2, R, 2 * C).double()
    inv2 = float(pl2.inv)
    Y2d = Y2.view(R, 2 * C).double()
    top2 = float(Y2d.abs().max())
    loose2 = 2.0 ** 15 / (top2 / inv2)
    err2 = ((h2[0] + h2[1] / 2048.0) * inv2 - Y2d).abs()
    assert 1.0 <= loose2 < 2.0 ** 8, loose2
    assert float(err2.max()) <= 2.0 ** -21 * top2 * loose2
    big2 = Y2d.abs() >= top2 * 2.0 ** -10
    assert float((err2[big2] / Y2d.abs()[big2]).max()) < 2.0 ** -21



@pytest.mark.parametrize("N,S,C,k", [(2, 8, 64, 5), (3, 16, 48, 3), (2, 16, 32, 5), (2, 32, 16, 5), (2, 32, 32, 3), (4, 8, 80, 3)])
def test_rfft2_also_computes_the_depthwise_conv(N, S, C, k):
    """ud_rfft2_ex_planes with dw_k (round 5): the kernel that transforms act(bn(x)) also writes the stride-1 depthwise conv of the
    same activated plane (SFConv's spatial branch) — against F.conv2d in float64 on the activation ud_rfft2_ex materialises, and
    the planes / activated output unchanged by the addition."""
    import torch.nn.functional as F
    from unidefense_amd import kernels as K
    dev = _dev()
    K.reset_zero_pool()
    g = torch.Generator().manual_seed(S * 100 + C + k)
    x = torch.randn(N, S, S, C, generator=g).to(dev)
    w = (0.3 * torch.randn(C, 1, k, k, generator=g))
    wt = w.view(C, k * k).t().contiguous().to(dev)
    gamma, beta = (1.0 + 0.3 * torch.randn(C, generator=g)).to(dev), (0.2 * torch.randn(C, generator=g)).to(dev)
    acc = K.zeros64(2 * C, x)
    K.colstats(x.view(-1, C), acc)
    bn = K.DeferredBN(acc, C, N * S * S, gamma, beta, 1e-3, 1)
    pl0, a0 = K.rfft2_ex_planes(x, 1.0 / S, 1.0, bn=bn, want_act=True)
    pl1, a1, spat = K.rfft2_ex_planes(x, 1.0 / S, 1.0, bn=bn, want_act=True, dw_wt=wt, dw_k=k)
    torch.cuda.synchronize()
    R = N * S * (S // 2 + 1)
    valid = lambda pl: pl.buf.view(2, pl.npanel, pl.panel // 32, 32)[:, :, :R]          # (rows up to the next multiple of 128 are slack)
    assert torch.equal(a0, a1) and torch.equal(valid(pl0), valid(pl1)) and torch.equal(pl0.inv, pl1.inv)
    ref = F.conv2d(a0.double().cpu().permute(0, 3, 1, 2), w.double(), padding=(k - 1) // 2, groups=C).permute(0, 2, 3, 1)
    err = float((spat.double().cpu() - ref).abs().max()) / float(ref.abs().max())
    assert err < 2e-6, err
