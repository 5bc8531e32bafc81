"""GPU parity tests, operator level: every HIP operator (forward AND backward, called through the C ABI
via unidefense_amd.tape) against the same op of the oracle / plain torch evaluated on the CPU in float64.

Tolerance: 1e-3 relative to the tensor's max magnitude (BASELINE.json north_star: "within 1e-3 rel fp32");
observed errors are ~1e-6..1e-5 and are printed.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

RTOL = 1e-3


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def to_pix(t):
    return t.permute(0, 2, 3, 1).contiguous()


def to_nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def check(name, got, ref, tol=RTOL):
    from tests.margins import within
    e = rel_err(got, ref)
    print(f"  {name}: rel err {e:.3e}")
    assert within(name, e, tol), f"{name}: rel err {e:.3e} > {tol}"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


def run_tape(fn, inputs, params, gouts_fn):
    """fn(tape, *inputs_cuda, *params_cuda) -> output or tuple.  Returns outputs, input grads, param grads."""
    from unidefense_amd import tape as T
    tape = T.Tape()
    outs = fn(tape, *inputs, *params)
    single = not isinstance(outs, (tuple, list))
    outs_l = [outs] if single else list(outs)
    gouts = gouts_fn(outs_l)
    for o, g in zip(outs_l, gouts):
        if g is not None:
            tape.add_grad(o, g)
    # collect input grads before replay clears them: wrap by recording a sentinel first node
    grads_in = {}
    pg = tape.param_grads

    def grab():
        for i, t in enumerate(inputs):
            grads_in[i] = tape.grads.get(id(t))
    tape.nodes.insert(0, grab)      # runs last in the reversed replay
    tape.backward()
    return outs_l, [grads_in.get(i) for i in range(len(inputs))], [pg.get(p) for p in params]


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 24), (1280, 3264, 3264 // 8), (520, 40, 96),
                                   (1000, 24, 144), (64, 3, 180), (33, 130, 20)])
def test_gemm_nt_nn_tn(M, N, K):
    dev = _dev()
    from unidefense_amd import kernels as Kk
    a, w = rnd(M, K, seed=1), rnd(N, K, seed=2)
    ref = a.double() @ w.double().t()
    check("nt", Kk.gemm_nt(a.to(dev), w.to(dev)), ref)
    wk = rnd(K, N, seed=3)
    check("nn", Kk.gemm_nn(a.to(dev), wk.to(dev)), a.double() @ wk.double())
    at, bt = rnd(K, M, seed=4), rnd(K, N, seed=5)
    check("tn", Kk.gemm_tn(at.to(dev), bt.to(dev)), at.double().t() @ bt.double())


@pytest.mark.parametrize("ci,co", [(20, 20), (20, 3), (3, 20), (3, 48), (32, 3), (3, 32)])
@pytest.mark.parametrize("geom", ["same", "stem_s2", "transposed_s2", "dgrad_of_convT"])
def test_conv_small_direct_kernel_equals_implicit_gemm(ci, co, geom):
    """csrc/conv_small.hip (one thread per output pixel, scalar-loaded weights) against F.conv2d / F.conv_transpose2d
    in float64 and against the implicit-GEMM path it replaces, for every geometry the decoder / stem use."""
    dev = _dev()
    from unidefense_amd import kernels as Kk
    g_ = torch.Generator().manual_seed(ci * 100 + co)
    n, h = 3, 80                                             # 3 * 80 * 80 = 19200 output pixels >= the dispatch threshold
    if geom == "same":
        x = torch.randn(n, ci, h, h, generator=g_)
        w = torch.randn(co, ci, 3, 3, generator=g_)
        ref = F.conv2d(x.double(), w.double(), padding=1)
        gm = Kk.conv_geom(n, h, h, ci, h, h, 3, 3, 1, 1, 1, 0)
        wmat = w.permute(0, 2, 3, 1).reshape(co, 9 * ci)
    elif geom == "stem_s2":                                  # static same padding (0,1,0,1), stride 2
        x = torch.randn(n, ci, 2 * h, 2 * h, generator=g_)
        w = torch.randn(co, ci, 3, 3, generator=g_)
        ref = F.conv2d(F.pad(x.double(), [0, 1, 0, 1]), w.double(), stride=2)
        gm = Kk.conv_geom(n, 2 * h, 2 * h, ci, h, h, 3, 3, 2, 0, 0, 0)
        wmat = w.permute(0, 2, 3, 1).reshape(co, 9 * ci)
    elif geom == "transposed_s2":                            # ConvTranspose2d(k3, s2, p1, op1): 40 -> 80
        x = torch.randn(n, ci, h // 2, h // 2, generator=g_)
        w = torch.randn(ci, co, 3, 3, generator=g_)
        ref = F.conv_transpose2d(x.double(), w.double(), stride=2, padding=1, output_padding=1)
        gm = Kk.conv_geom(n, h // 2, h // 2, ci, h, h, 3, 3, 2, 1, 1, 1)
        wmat = w.permute(1, 2, 3, 0).reshape(co, 9 * ci)
    else:                                                    # data gradient of that ConvTranspose2d: stride-2 conv over dY
        x = torch.randn(n, ci, 2 * h, 2 * h, generator=g_)  # dY with `ci` channels
        w = torch.randn(co, ci, 3, 3, generator=g_)         # ConvTranspose weight [Cin_T = co, Cout_T = ci]
        ref = F.conv2d(x.double(), w.double(), stride=2, padding=1)
        gm = Kk.conv_geom(n, 2 * h, 2 * h, ci, h, h, 3, 3, 2, 1, 1, 0)
        wmat = w.permute(0, 2, 3, 1).reshape(co, 9 * ci)
    xp, wm = to_pix(x).to(dev), wmat.contiguous().to(dev)
    assert Kk._conv_small_supported(ci, co, 3, 3)
    saved = Kk._CONV_SMALL, Kk._CONV_SMALL_MIN_M
    try:
        Kk._CONV_SMALL, Kk._CONV_SMALL_MIN_M = True, 1
        direct = Kk.conv_gather_nt(xp, wm, gm)
        Kk._CONV_SMALL = False
        gemm = Kk.conv_gather_nt(xp, wm, gm)
    finally:
        Kk._CONV_SMALL, Kk._CONV_SMALL_MIN_M = saved
    check(f"conv_small {geom} {ci}->{co} vs fp64", to_nchw(direct), ref, 1e-5)
    check(f"conv_small {geom} {ci}->{co} vs implicit GEMM", direct, gemm, 1e-5)


@pytest.mark.parametrize("ci,ma", [(20, 20), (20, 3), (3, 48), (3, 20), (32, 3)])
@pytest.mark.parametrize("geom", ["same", "stem_s2", "stride2_pad1"])
def test_conv_small_wgrad_equals_implicit_gemm(ci, ma, geom):
    """csrc/conv_small.hip weight gradient (LDS-staged rows, 4x4 register blocks, partials + sum) against the
    autograd weight gradient of F.conv2d in float64 and the split-K implicit GEMM it replaces; the row count is not a
    multiple of the tile (ragged last tile and last workgroup)."""
    dev = _dev()
    from unidefense_amd import kernels as Kk
    g_ = torch.Generator().manual_seed(ci * 100 + ma + 7)
    n, h = 3, 37
    if geom == "same":
        hin, stride, pt, pads = h, 1, 1, [1, 1, 1, 1]
    elif geom == "stem_s2":
        hin, stride, pt, pads = 2 * h, 2, 0, [0, 1, 0, 1]
    else:
        hin, stride, pt, pads = 2 * h, 2, 1, [1, 1, 1, 1]
    x = torch.randn(n, ci, hin, hin, generator=g_, dtype=torch.float64)
    w = torch.zeros(ma, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(n, ma, h, h, generator=g_, dtype=torch.float64)
    y = F.conv2d(F.pad(x, pads), w, stride=stride)
    assert y.shape == dy.shape
    (ref,) = torch.autograd.grad(y, w, dy)                                  # [ma, ci, 3, 3]
    ref = ref.permute(0, 2, 3, 1).reshape(ma, 9 * ci)
    gm = Kk.conv_geom(n, hin, hin, ci, h, h, 3, 3, stride, pt, pt, 0)
    a = to_pix(dy.float()).reshape(-1, ma).to(dev)
    xp = to_pix(x.float()).to(dev)
    saved = Kk._CONV_SMALL, Kk._CONV_SMALL_MIN_M
    try:
        Kk._CONV_SMALL, Kk._CONV_SMALL_MIN_M = True, 1
        direct = Kk.conv_gather_wgrad(a, xp, gm)
        Kk._CONV_SMALL = False
        gemm = Kk.conv_gather_wgrad(a, xp, gm)
    finally:
        Kk._CONV_SMALL, Kk._CONV_SMALL_MIN_M = saved
    check(f"conv_small wgrad {geom} {ci}x{ma} vs fp64", direct, ref, 1e-5)
    check(f"conv_small wgrad {geom} {ci}x{ma} vs implicit GEMM", direct, gemm, 1e-5)


@pytest.mark.parametrize("kind", ["nt", "nn"])
def test_gemm_tail_split_plan(kind):
    """kernels._tail_plan: 36 x 15 = 540 tiles -> 34 row-tiles in one plain launch + 2 row-tiles split-K (atomic
    accumulation into the zeroed tail rows); the assembled result must be the plain GEMM."""
    dev = _dev()
    from unidefense_amd import kernels as Kk
    M, N, K = 4608, 1920, 768
    assert Kk._tail_plan(M, N, K) == (4352, 3)
    g = torch.Generator().manual_seed(11)
    a = torch.randn(M, K, generator=g)
    w = torch.randn((N, K) if kind == "nt" else (K, N), generator=g)
    ref = a.double() @ (w.double().t() if kind == "nt" else w.double())
    got = (Kk.gemm_nt if kind == "nt" else Kk.gemm_nn)(a.to(dev), w.to(dev))
    check(f"tail-split {kind}", got, ref, 1e-5)


@pytest.mark.parametrize("kind,M,N,K", [("nt", 1280, 3264, 3264), ("nn", 1152, 256, 256), ("tn", 672, 672, 17408),
                                        ("nt", 200, 72, 100), ("nn", 132, 68, 36), ("tn", 68, 76, 1001)])
def test_gemm_split_bf16_path_has_fp32_accuracy(kind, M, N, K):
    """csrc/gemm_x3.hip (three exact bf16 pieces per operand, six products on the BF16 matrix pipe) against the
    v_mfma_f32_32x32x2_f32 kernel and float64: the error relative to sum|a||b| must not exceed the fp32 kernel's
    (x1.5 + 1e-7 slack), on wide-dynamic-range random operands and on same-sign operands (linearly growing sums)."""
    dev = _dev()
    from unidefense_amd import kernels as Kk, lib
    g = torch.Generator().manual_seed(7)
    for same_sign in (False, True):
        sa, sb = ((K, M) if kind == "tn" else (M, K)), ((N, K) if kind == "nt" else (K, N))
        a, b = torch.randn(sa, generator=g), torch.randn(sb, generator=g)
        a, b = a * torch.exp(2 * torch.randn(sa, generator=g)), b * torch.exp(2 * torch.randn(sb, generator=g))
        if same_sign:
            a, b = a.abs(), b.abs()
        A = a.double().t() if kind == "tn" else a.double()
        B = b.double().t() if kind == "nt" else b.double()
        ref, scale = A @ B, A.abs() @ B.abs()
        errs = []
        try:
            for path in (1, 2):
                lib.call("ud_gemm_set_path", path)
                y = {"nt": Kk.gemm_nt, "nn": Kk.gemm_nn, "tn": Kk.gemm_tn}[kind](a.to(dev), b.to(dev))
                errs.append(((y.double().cpu() - ref).abs() / scale).max().item())
        finally:
            lib.call("ud_gemm_set_path", 0)
        print(f"  {kind} {M}x{N}x{K} same_sign={same_sign}: fp32-mfma {errs[0]:.3e}  split-bf16 {errs[1]:.3e}")
        assert errs[1] <= 1.5 * errs[0] + 1e-7, errs
        assert errs[1] <= 2e-5


@pytest.mark.parametrize("kind,M,N,K", [("nt", 2048, 272, 1632), ("nn", 2048, 1632, 272), ("tn", 1632, 272, 2048),
                                        ("nt", 300, 72, 200), ("nn", 132, 68, 260), ("tn", 68, 76, 1001),
                                        ("nt", 8192 + 64, 160, 960), ("tn", 160, 960, 8192 + 3)])
@pytest.mark.parametrize("mode", ["ordered-slices", "atomics"])
def test_gemm_every_tile_configuration_and_split(kind, M, N, K, mode):
    """Every plan the per-shape tuner (kernels._tuned_plan) may pick: the four tiles of gemm_x3.hip (128x128, 128x64,
    64x128, 64x64, each with its own prefetch depth) x split-K 1 ... 24, on ragged shapes, against float64 — a plan is a
    speed choice, never a numerical one.  Both forms of split-K: partial products into ordered slices + ud_sum_slices
    (cfg.deterministic, the default: also required to be BITWISE repeatable) and fp32 atomics."""
    dev = _dev()
    from unidefense_amd import kernels as Kk
    from unidefense_amd.config import override
    with override(deterministic=mode == "ordered-slices"):
        _every_tile_and_split(dev, Kk, kind, M, N, K, mode)


def _every_tile_and_split(dev, Kk, kind, M, N, K, mode):
    from tests.margins import within
    g = torch.Generator().manual_seed(M + N)
    sa, sb = ((K, M) if kind == "tn" else (M, K)), ((N, K) if kind == "nt" else (K, N))
    a, b = torch.randn(sa, generator=g), torch.randn(sb, generator=g)
    A = a.double().t() if kind == "tn" else a.double()
    B = b.double().t() if kind == "nt" else b.double()
    ref, scale = A @ B, A.abs() @ B.abs()
    a, b = a.to(dev), b.to(dev)
    a_mode, b_mode = (1, 1) if kind == "tn" else (0, 0 if kind == "nt" else 1)
    lda, ldb = (M if kind == "tn" else K), (K if kind == "nt" else N)
    worst = 0.0
    for cfg in (1, 2, 3, 4):
        for split in (1, 2, 3, 8, 24):
            if split > 1 and K // split < 16:
                continue
            out = torch.zeros(M, N, device=dev)
            Kk._gemm(a, b, out, M, N, K, lda, ldb, N, a_mode, b_mode, 2 if split > 1 else 0, split, cfg=cfg)
            e = ((out.double().cpu() - ref).abs() / scale).max().item()
            worst = max(worst, e)
            assert e <= 2e-6, (cfg, split, e)
            if split > 1 and mode == "ordered-slices":
                # a fresh (uninitialised) result buffer takes the write form of the slice sum; twice the same bits
                o1, o2 = Kk.split_out((M, N), a), Kk.split_out((M, N), a)
                o1.fill_(float("nan"))
                Kk._gemm(a, b, o1, M, N, K, lda, ldb, N, a_mode, b_mode, 2, split, cfg=cfg)
                Kk._gemm(a, b, o2, M, N, K, lda, ldb, N, a_mode, b_mode, 2, split, cfg=cfg)
                assert torch.equal(o1, out) and torch.equal(o2, out), (cfg, split)
    within(f"{kind} {M}x{N}x{K} [{mode}]: worst error / sum|a||b| over 4 tiles x 5 splits", worst, 2e-6)
    print(f"  {kind} {M}x{N}x{K} [{mode}]: worst error / sum|a||b| over 4 tiles x 5 splits: {worst:.2e}")


def test_syncbn_combine_matches_gloo_tested_formula():
    """ud_syncbn_combine (the product's SyncBatchNorm fold) against tape.sync_batch_stats' math — the function the
    world_size-2 gloo test (tests/test_parallel_cpu.py) checks against torch.nn.SyncBatchNorm semantics."""
    dev = _dev()
    from unidefense_amd import kernels as Kk
    world, C, rows = 4, 272, 2048
    full = rnd(world * rows, C, seed=3) * 2.0 + 0.7
    shards = full.view(world, rows, C)
    gathered = torch.stack([torch.stack([s.mean(0), s.var(0, unbiased=False)]) for s in shards])      # [world, 2, C]
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    mean, invstd = Kk.syncbn_combine(gathered.to(dev).contiguous(), world, C, rows, 1e-3, 0.1, rm, rv)
    ref_mean = full.double().mean(0)
    ref_var = full.double().var(0, unbiased=False)
    check("syncbn mean", mean.view(-1), ref_mean, 1e-5)
    check("syncbn invstd", invstd.view(-1), 1.0 / torch.sqrt(ref_var + 1e-3), 1e-5)
    n = world * rows
    check("syncbn running_mean", rm, 0.1 * ref_mean, 1e-5)
    check("syncbn running_var", rv, 0.9 + 0.1 * ref_var * n / (n - 1), 1e-5)


@pytest.mark.parametrize("N,D,R", [(32, 160, 16), (8, 448, 4), (64, 1792, 32), (6, 40, 3)])
def test_fused_aw_triplet_value_and_gradient(N, D, R):
    """ud_aw_triplet (loss + d loss/d feat in two launches) against the oracle's restatement of
    loss/triplet_loss.py in float64, through the loss module the engine uses (n_real hint set)."""
    dev = _dev()
    from oracle import losses as OL
    from unidefense_amd.loss.triplet_loss import AsymmetricalWeightedTripletLoss
    feat = rnd(N, D, seed=11) * 0.3
    labels = torch.tensor([0] * R + [1] * (N - R))
    fr = feat.double().requires_grad_()
    ref = OL.aw_triplet(fr, labels)
    ref.backward()
    crit = AsymmetricalWeightedTripletLoss()
    crit.n_real = R
    f = feat.to(dev).requires_grad_()
    loss = crit(f, labels.to(dev))
    (3.0 * loss).backward()
    check("aw_triplet loss", loss.reshape(1), ref.detach().reshape(1), 1e-5)
    check("aw_triplet dfeat", f.grad / 3.0, fr.grad, 1e-4)


@pytest.mark.parametrize("world,R,C,act", [(2, 2048, 272, 1), (4, 512, 24, 0), (8, 64, 1792, 1)])
def test_syncbn_shards_equal_full_batch(world, R, C, act):
    """The SyncBatchNorm data path of tape.batchnorm_act on ONE GPU: the batch is cut into `world` shards, every
    shard runs the per-rank kernels (norm_stats_local, norm_apply, norm_bwd_sums, norm_bwd_apply) and the collectives
    are emulated in place (all_gather = stack, all_reduce = sum).  Must equal plain BatchNorm over the whole batch,
    forward and backward — the N > 1 arithmetic that only the driver's multi-GPU run would otherwise exercise."""
    dev = _dev()
    from unidefense_amd import kernels as Kk
    eps = 1e-3
    x = (rnd(world * R, C, seed=1) * 1.5 + 0.3).to(dev)
    dy = rnd(world * R, C, seed=2).to(dev)
    g, b = (rnd(C, seed=3) * 0.1 + 1).to(dev), (rnd(C, seed=4) * 0.1).to(dev)
    mean, invstd = Kk.norm_stats(x, 1, world * R, eps)
    y_ref = Kk.norm_apply(x, 1, world * R, mean, invstd, g, b, act)
    dx_ref, dg_ref, db_ref = Kk.norm_bwd(x, dy, 1, world * R, mean, invstd, g, b, act)
    xs, dys = x.view(world, R, C), dy.view(world, R, C)
    gathered = torch.stack([Kk.norm_stats_local(xs[r].contiguous(), 1, R, eps).view(2, C) for r in range(world)])
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    mean2, invstd2 = Kk.syncbn_combine(gathered.contiguous(), world, C, R, eps, 0.1, rm, rv)
    check("syncbn mean", mean2, mean, 1e-5)
    check("syncbn invstd", invstd2, invstd, 1e-5)
    y = torch.cat([Kk.norm_apply(xs[r].contiguous(), 1, R, mean2, invstd2, g, b, act) for r in range(world)])
    check("syncbn y", y, y_ref, 1e-5)
    sums = [Kk.norm_bwd_sums(xs[r].contiguous(), dys[r].contiguous(), 1, R, mean2, invstd2, g, b, act) for r in range(world)]
    s = sum(t[0] for t in sums)                                   # all_reduce of [2, 1, C]
    dx = torch.cat([Kk.norm_bwd_apply(xs[r].contiguous(), dys[r].contiguous(), 1, R, mean2, invstd2, g, b, s,
                                      1.0 / (world * R), act) for r in range(world)])
    check("syncbn dx", dx, dx_ref, 2e-5)
    check("syncbn dgamma", sum(t[1] for t in sums), dg_ref, 2e-5)
    check("syncbn dbeta", sum(t[2] for t in sums), db_ref, 2e-5)


@pytest.mark.parametrize("Hi,Ho", [(7, 19), (16, 5), (128, 256), (1, 4), (9, 9), (12, 1)])
def test_bilinear_backward_gather(Hi, Ho):
    """ud_bilinear_bwd (gather over the output pixels that interpolate from each input pixel) vs torch autograd of
    F.interpolate(mode='bilinear', align_corners=True), up- and down-sampling, degenerate sizes."""
    dev = _dev()
    from unidefense_amd import kernels as Kk
    Wi, Wo = Hi + 3, Ho + 2
    x = rnd(2, 3, Hi, Wi, seed=1).double().requires_grad_()
    y = F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=True)
    gy = rnd(2, 3, Ho, Wo, seed=2)
    y.backward(gy.double())
    dx = Kk.bilinear_bwd(gy.to(dev), Hi, Wi)
    # 1e-4: the source coordinate o*(in-1)/(out-1) is evaluated in fp32 (torch: same formula, other rounding)
    check(f"bilinear bwd {Hi}->{Ho}", dx, x.grad, 1e-4)
    check(f"bilinear fwd {Hi}->{Ho}", Kk.bilinear_fwd(x.detach().float().to(dev), Ho, Wo), y.detach(), 1e-4)


def test_gemm_tn_splitk_and_accumulate():
    dev = _dev()
    from unidefense_amd import kernels as Kk
    K_, M, N = 70000, 96, 48
    at, bt = rnd(K_, M, seed=4), rnd(K_, N, seed=5)
    check("tn split-k", Kk.gemm_tn(at.to(dev), bt.to(dev)), at.double().t() @ bt.double())
    a, w = rnd(200, 64, seed=1), rnd(72, 64, seed=2)
    base = rnd(200, 72, seed=3)
    out = base.to(dev).clone()
    Kk.gemm_nt(a.to(dev), w.to(dev), out=out, accumulate=True)
    check("nt accumulate", out, base.double() + a.double() @ w.double().t())


@pytest.mark.parametrize("N,H,Ci,Co", [(2, 16, 160, 80), (2, 8, 272, 272), (1, 32, 20, 3), (2, 16, 40, 20)])
def test_conv3x3(N, H, Ci, Co):
    dev = _dev()
    from unidefense_amd import tape as T
    x = rnd(N, Ci, H, H, seed=1)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=0.1)
    gy = rnd(N, Co, H, H, seed=3)
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    yr = F.conv2d(xr, wr, None, 1, 1)
    yr.backward(gy.double())
    xp, wp = to_pix(x).to(dev), w.to(dev)
    outs, gin, gp = run_tape(lambda t, a, b: T.conv_dense(t, a, b, 1, 1, 1, H, H), [xp], [wp],
                             lambda o: [to_pix(gy).to(dev)])
    check("y", to_nchw(outs[0]), yr)
    check("dx", to_nchw(gin[0]), xr.grad)
    check("dw", gp[0], wr.grad)


def test_stem_conv():
    dev = _dev()
    from unidefense_amd import tape as T
    N, H = 2, 64
    x = rnd(N, 3, H, H, seed=1)
    w = rnd(48, 3, 3, 3, seed=2, scale=0.3)
    gy = rnd(N, 48, H // 2, H // 2, seed=3)
    xr, wr = x.double(), w.double().requires_grad_()
    yr = F.conv2d(F.pad(xr, [0, 1, 0, 1]), wr, None, 2, 0)
    yr.backward(gy.double())
    outs, gin, gp = run_tape(lambda t, a, b: T.conv_dense(t, a, b, 2, 0, 0, H // 2, H // 2, need_dx=False),
                             [to_pix(x).to(dev)], [w.to(dev)], lambda o: [to_pix(gy).to(dev)])
    check("y", to_nchw(outs[0]), yr)
    check("dw", gp[0], wr.grad)


@pytest.mark.parametrize("N,H,C", [(2, 16, 80), (1, 32, 20)])
def test_conv_transpose(N, H, C):
    dev = _dev()
    from unidefense_amd import tape as T
    x = rnd(N, C, H, H, seed=1)
    w = rnd(C, C, 3, 3, seed=2, scale=0.1)
    gy = rnd(N, C, 2 * H, 2 * H, seed=3)
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    yr = F.conv_transpose2d(xr, wr, None, 2, 1, 1)
    yr.backward(gy.double())
    outs, gin, gp = run_tape(lambda t, a, b: T.conv_transpose_s2(t, a, b), [to_pix(x).to(dev)], [w.to(dev)],
                             lambda o: [to_pix(gy).to(dev)])
    check("y", to_nchw(outs[0]), yr)
    check("dx", to_nchw(gin[0]), xr.grad)
    check("dw", gp[0], wr.grad)


@pytest.mark.parametrize("k,s,pad,H,C", [(3, 1, (1, 1, 1, 1), 16, 48), (3, 2, (0, 1, 0, 1), 32, 144),
                                         (5, 1, (2, 2, 2, 2), 16, 960), (5, 2, (2, 2, 2, 2), 32, 192),
                                         (5, 2, (1, 2, 1, 2), 16, 960), (3, 1, (1, 1, 1, 1), 8, 2688)])
def test_dwconv(k, s, pad, H, C):
    dev = _dev()
    from unidefense_amd import tape as T
    N = 2
    x = rnd(N, C, H, H, seed=1)
    w = rnd(C, 1, k, k, seed=2, scale=0.3)
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    yr = F.conv2d(F.pad(xr, list(pad)), wr, None, s, 0, 1, C)
    gy = rnd(*yr.shape, seed=3)
    yr.backward(gy.double())
    outs, gin, gp = run_tape(lambda t, a, b: T.dwconv(t, a, b, s, pad), [to_pix(x).to(dev)], [w.to(dev)],
                             lambda o: [to_pix(gy).to(dev)])
    check("y", to_nchw(outs[0]), yr)
    check("dx", to_nchw(gin[0]), xr.grad)
    check("dw", gp[0], wr.grad)


@pytest.mark.parametrize("N,H,C,act", [(4, 16, 144, 1), (2, 8, 1632, 0), (3, 32, 24, 1), (2, 4, 2688, 1)])
def test_batchnorm(N, H, C, act):
    dev = _dev()
    from unidefense_amd import tape as T
    x = rnd(N, C, H, H, seed=1) * 2 + 0.5
    g, b = rnd(C, seed=2) * 0.1 + 1, rnd(C, seed=3) * 0.1
    rm, rv = torch.zeros(C), torch.ones(C)
    xr, gr, br = x.double().requires_grad_(), g.double().requires_grad_(), b.double().requires_grad_()
    rm_r, rv_r = rm.double().clone(), rv.double().clone()
    yr = F.batch_norm(xr, rm_r, rv_r, gr, br, True, 0.01, 1e-3)
    if act:
        yr = yr * torch.sigmoid(yr)
    gy = rnd(*yr.shape, seed=4)
    yr.backward(gy.double())
    rm_d, rv_d = rm.to(dev), rv.to(dev)
    outs, gin, gp = run_tape(
        lambda t, a, w_, b_: T.batchnorm_act(t, a, w_, b_, rm_d, rv_d, 1e-3, 0.01, True, act),
        [to_pix(x).to(dev)], [g.to(dev).requires_grad_(), b.to(dev).requires_grad_()], lambda o: [to_pix(gy).to(dev)])
    check("y", to_nchw(outs[0]), yr)
    check("dx", to_nchw(gin[0]), xr.grad)
    check("dgamma", gp[0], gr.grad)
    check("dbeta", gp[1], br.grad)
    check("running_mean", rm_d, rm_r)
    check("running_var", rv_d, rv_r)


def test_batchnorm1d_rows():
    dev = _dev()
    from unidefense_amd import tape as T
    N, C = 32, 1792
    x = rnd(N, C, seed=1) + 1.0
    g, b = rnd(C, seed=2) * 0.1 + 1, rnd(C, seed=3) * 0.1
    xr, gr = x.double().requires_grad_(), g.double().requires_grad_()
    yr = F.batch_norm(xr, None, None, gr, b.double(), True, 0.1, 1e-5)
    gy = rnd(N, C, seed=4)
    yr.backward(gy.double())
    rm_d, rv_d = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    outs, gin, gp = run_tape(lambda t, a, w_, b_: T.batchnorm_act(t, a, w_, b_, rm_d, rv_d, 1e-5, 0.1, True, 0),
                             [x.to(dev)], [g.to(dev).requires_grad_(), b.to(dev)], lambda o: [gy.to(dev)])
    check("y", outs[0], yr)
    check("dx", gin[0], xr.grad)
    check("dgamma", gp[0], gr.grad)


@pytest.mark.parametrize("N,H,C", [(2, 32, 80), (3, 16, 20), (2, 128, 20)])
def test_instancenorm_swish(N, H, C):
    dev = _dev()
    from unidefense_amd import tape as T
    x = rnd(N, C, H, H, seed=1) * 1.5 + 0.3
    g, b = rnd(C, seed=2) * 0.1 + 1, rnd(C, seed=3) * 0.1
    xr, gr, br = x.double().requires_grad_(), g.double().requires_grad_(), b.double().requires_grad_()
    yr = F.instance_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
    yr = yr * torch.sigmoid(yr)
    gy = rnd(*yr.shape, seed=4)
    yr.backward(gy.double())
    outs, gin, gp = run_tape(lambda t, a, w_, b_: T.instancenorm_act(t, a, w_, b_, 1e-5, 1), [to_pix(x).to(dev)],
                             [g.to(dev).requires_grad_(), b.to(dev).requires_grad_()], lambda o: [to_pix(gy).to(dev)])
    check("y", to_nchw(outs[0]), yr)
    check("dx", to_nchw(gin[0]), xr.grad)
    check("dgamma", gp[0], gr.grad)
    check("dbeta", gp[1], br.grad)


@pytest.mark.parametrize("S,C,norm", [(8, 1632, "ortho"), (8, 3, "ortho"), (8, 272, "ortho"), (16, 672, "ortho"),
                                      (16, 40, None), (32, 336, "ortho"), (64, 192, "ortho"), (32, 20, None),
                                      # 5 * 2^k sizes (ResNet50 variant at 320 x 320): mixed-radix in-register DFT
                                      (10, 512, "ortho"), (10, 3, None), (20, 256, "ortho"), (40, 128, None),
                                      (80, 128, "ortho"), (80, 7, "ortho"),
                                      # 3 * 2^k sizes (EfficientNet-b4 at its native 380 x 380: maps of 48 / 24 / 12)
                                      (12, 1632, "ortho"), (12, 272, None), (24, 672, "ortho"), (24, 960, "ortho"),
                                      (48, 336, "ortho"), (48, 5, None),
                                      # any other side: DFT matrices on the GEMM kernels (95 = 5 * 19: the 380 x 380 trunk's first SF block)
                                      # (two batched GEMMs on the pixel-major tensor; 13: matrix rows padded to 4; C = 6: the plane-copy form)
                                      (95, 192, "ortho"), (95, 8, None), (14, 24, "ortho"), (13, 12, None), (14, 6, "ortho")])
def test_rfft2_irfft2(S, C, norm):
    dev = _dev()
    from unidefense_amd import tape as T
    N = 2
    x = rnd(N, C, S, S, seed=1)
    xr = x.double().requires_grad_()
    fr = torch.fft.rfft2(xr, norm=norm)
    yr = torch.cat([fr.real, fr.imag], 1)
    gy = rnd(*yr.shape, seed=2)
    yr.backward(gy.double())
    outs, gin, _ = run_tape(lambda t, a: T.rfft2_cat(t, a, norm), [to_pix(x).to(dev)], [],
                            lambda o: [to_pix(gy).to(dev)])
    check("rfft2", to_nchw(outs[0]), yr)
    check("rfft2 adjoint", to_nchw(gin[0]), xr.grad)
    # inverse
    y = rnd(N, 2 * C, S, S // 2 + 1, seed=3)
    yr2 = y.double().requires_grad_()
    re, im = torch.tensor_split(yr2, 2, dim=1)
    xr2 = torch.fft.irfft2(torch.complex(re, im), s=(S, S), norm=norm)
    gx = rnd(N, C, S, S, seed=4)
    xr2.backward(gx.double())
    outs, gin, _ = run_tape(lambda t, a: T.irfft2_split(t, a, norm), [to_pix(y).to(dev)], [],
                            lambda o: [to_pix(gx).to(dev)])
    check("irfft2", to_nchw(outs[0]), xr2)
    check("irfft2 adjoint", to_nchw(gin[0]), yr2.grad)


@pytest.mark.parametrize("C,k,s,pad,S", [(192, 5, 2, (2, 2, 2, 2), 64), (336, 5, 1, (2, 2, 2, 2), 32),
                                         (672, 3, 1, (1, 1, 1, 1), 16), (960, 5, 2, (1, 2, 1, 2), 16),
                                         (1632, 5, 1, (2, 2, 2, 2), 8)])
def test_sfconv(C, k, s, pad, S):
    """SFConv2dStaticSamePadding.forward/backward vs the oracle restatement (oracle/eb4.py:sfconv)."""
    dev = _dev()
    from unidefense_amd import tape as T
    from oracle import eb4
    N = 2
    x = rnd(N, C, S, S, seed=1)
    w = rnd(C, 1, k, k, seed=2, scale=0.3)
    wf = rnd(2 * C, 2 * C, 1, 1, seed=3, scale=math.sqrt(1.0 / C))
    alpha = torch.tensor(0.2)
    sd = {"p.weight": w.double().requires_grad_(), "p.freq_conv.weight": wf.double().requires_grad_(),
          "p.sf_coef": alpha.double().requires_grad_()}
    xr = x.double().requires_grad_()
    yr = eb4.sfconv(xr, sd, "p", s, pad, "ortho")
    gy = rnd(*yr.shape, seed=4)
    yr.backward(gy.double())
    outs, gin, gp = run_tape(lambda t, a, w_, wf_, al: T.sfconv_dw(t, a, w_, wf_, al, s, pad, "ortho"),
                             [to_pix(x).to(dev)], [w.to(dev), wf.to(dev), alpha.to(dev)],
                             lambda o: [to_pix(gy).to(dev)])
    check("y", to_nchw(outs[0]), yr)
    check("dx", to_nchw(gin[0]), xr.grad)
    check("dw", gp[0], sd["p.weight"].grad)
    check("dwf", gp[1], sd["p.freq_conv.weight"].grad)
    check("dalpha", gp[2].reshape(()), sd["p.sf_coef"].grad)


@pytest.mark.parametrize("C,Cs,H", [(144, 6, 32), (1632, 68, 8), (336, 14, 16)])
def test_squeeze_excite(C, Cs, H):
    dev = _dev()
    from unidefense_amd import tape as T
    N = 3
    x = rnd(N, C, H, H, seed=1)
    wr_, br_ = rnd(Cs, C, 1, 1, seed=2, scale=0.1), rnd(Cs, seed=3, scale=0.1)
    we_, be_ = rnd(C, Cs, 1, 1, seed=4, scale=0.3), rnd(C, seed=5, scale=0.1)
    ps = [t.double().requires_grad_() for t in (wr_, br_, we_, be_)]
    xr = x.double().requires_grad_()
    s = F.adaptive_avg_pool2d(xr, 1)
    s = F.conv2d(s, ps[0], ps[1])
    s = s * torch.sigmoid(s)
    s = F.conv2d(s, ps[2], ps[3])
    yr = torch.sigmoid(s) * xr
    gy = rnd(*yr.shape, seed=6)
    yr.backward(gy.double())
    outs, gin, gp = run_tape(lambda t, a, *p: T.squeeze_excite(t, a, *p), [to_pix(x).to(dev)],
                             [t.to(dev) for t in (wr_, br_, we_, be_)], lambda o: [to_pix(gy).to(dev)])
    check("y", to_nchw(outs[0]), yr)
    check("dx", to_nchw(gin[0]), xr.grad)
    for i, nm in enumerate(("dWr", "dbr", "dWe", "dbe")):
        check(nm, gp[i], ps[i].grad)


def test_small_ops():
    dev = _dev()
    from unidefense_amd import tape as T
    N, C, H = 4, 160, 16
    x, skip = rnd(N, C, H, H, seed=1), rnd(N, C, H, H, seed=2)
    keep = torch.tensor([1.0, 0.0, 1.0, 1.0])
    gy = rnd(N, C, H, H, seed=3)
    # residual + drop connect
    xr, sr = x.double().requires_grad_(), skip.double().requires_grad_()
    yr = xr / 0.9 * keep.double().view(-1, 1, 1, 1) + sr
    yr.backward(gy.double())
    outs, gin, _ = run_tape(lambda t, a, b: T.residual(t, a, b, keep.to(dev), 0.9),
                            [to_pix(x).to(dev), to_pix(skip).to(dev)], [], lambda o: [to_pix(gy).to(dev)])
    check("residual y", to_nchw(outs[0]), yr)
    check("residual dx", to_nchw(gin[0]), xr.grad)
    check("residual dskip", to_nchw(gin[1]), sr.grad)
    # mean over HW
    xr = x.double().requires_grad_()
    mr = xr.mean((2, 3))
    gm = rnd(N, C, seed=4)
    mr.backward(gm.double())
    outs, gin, _ = run_tape(lambda t, a: T.mean_hw(t, a), [to_pix(x).to(dev)], [], lambda o: [gm.to(dev)])
    check("mean_hw", outs[0], mr)
    check("mean_hw dx", to_nchw(gin[0]), xr.grad)
    # gate mix
    al = torch.tensor(0.3)
    pr, qr, ar = x.double().requires_grad_(), skip.double().requires_grad_(), al.double().requires_grad_()
    a_ = torch.sigmoid(ar)
    yr = (1 - a_) * pr + a_ * qr
    yr.backward(gy.double())
    outs, gin, gp = run_tape(lambda t, a, b, c: T.gate_mix(t, a, b, c), [x.to(dev), skip.to(dev)], [al.to(dev)],
                             lambda o: [gy.to(dev)])
    check("gate_mix y", outs[0], yr)
    check("gate_mix dp", gin[0], pr.grad)
    check("gate_mix dq", gin[1], qr.grad)
    check("gate_mix dalpha", gp[0].reshape(()), ar.grad)
    # dropout with mask, linear
    mask = (rnd(N, 1792, seed=5) > 0).float()
    f = rnd(N, 1792, seed=6)
    w, b = rnd(2, 1792, seed=7, scale=0.05), rnd(2, seed=8)
    fr, wr, br = f.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    yr = F.linear(fr * mask.double() / 0.5, wr, br)
    gl = rnd(N, 2, seed=9)
    yr.backward(gl.double())
    outs, gin, gp = run_tape(lambda t, a, w_, b_: T.linear(t, T.dropout_mask(t, a, mask.to(dev), 0.5), w_, b_),
                             [f.to(dev)], [w.to(dev), b.to(dev)], lambda o: [gl.to(dev)])
    check("linear y", outs[0], yr)
    check("linear dx", gin[0], fr.grad)
    check("linear dW", gp[0], wr.grad)
    check("linear db", gp[1], br.grad)


def test_image_tail():
    """tanh -> planes -> bilinear(align_corners) -> L1 + frequency L1 (model/unidefense.py:101,244-253)."""
    dev = _dev()
    from unidefense_amd import tape as T
    N, S = 2, 64
    d = rnd(N, 3, S, S, seed=1)
    x = (torch.rand(N, 3, 2 * S, 2 * S, generator=torch.Generator().manual_seed(2)) * 2 - 1).float()
    dr = d.double().requires_grad_()
    rec_r = F.interpolate(torch.tanh(dr), size=(2 * S, 2 * S), mode="bilinear", align_corners=True)
    sp_r = (rec_r - x.double()).abs().mean((1, 2, 3))
    fa = torch.fft.rfft2(rec_r, norm="ortho")
    fb = torch.fft.rfft2(x.double(), norm="ortho")
    fq_r = ((fa.real - fb.real).abs() + (fa.imag - fb.imag).abs()).mean((1, 2, 3))
    gs, gf = rnd(N, seed=3), rnd(N, seed=4)
    grec = rnd(N, 3, 2 * S, 2 * S, seed=5) * 1e-4
    (sp_r * gs.double()).sum().add((fq_r * gf.double()).sum()).add((rec_r * grec.double()).sum()).backward()
    xd = x.to(dev)

    def fn(t, a):
        planes = T.tanh_to_planes(t, a)
        rec = T.bilinear(t, planes, 2 * S, 2 * S)
        sp, fq = T.rec_losses(t, rec, xd, "ortho")
        return rec, sp, fq
    outs, gin, _ = run_tape(fn, [to_pix(d).to(dev)], [], lambda o: [grec.to(dev), gs.to(dev), gf.to(dev)])
    check("rec", outs[0], rec_r)
    check("spatial", outs[1], sp_r)
    check("freq", outs[2], fq_r)
    check("d(dec3)", to_nchw(gin[0]), dr.grad)


def test_bilinear_down():
    dev = _dev()
    from unidefense_amd import kernels as Kk
    x = rnd(2, 3, 128, 128, seed=1)
    ref = F.interpolate(x.double(), size=(8, 8), mode="bilinear", align_corners=True)
    check("128->8", Kk.bilinear_fwd(x.to(dev), 8, 8), ref)


@pytest.mark.parametrize("kind", ["freq", "spat"])
def test_dynamic_filter(kind):
    """FrequencyDynamicFilter / SpatialDynamicFilter (model/modules.py:79-134) vs oracle.dynamic_filter."""
    dev = _dev()
    from unidefense_amd import tape as T
    from oracle import eb4
    N, h = 4, 8
    if kind == "freq":
        C, w_, D, k = 544, 5, 6, 1
    else:
        C, w_, D, k = 272, 8, 3, 3
    x = rnd(N, C, h, w_, seed=1)
    diff = rnd(N, D, h, w_, seed=2).abs()
    sd = {"f.layer1.0.weight": rnd(C, C, k, k, seed=3, scale=math.sqrt(2.0 / (C * k * k))).double().requires_grad_(),
          "f.layer1.1.weight": (rnd(C, seed=4) * 0.1 + 1).double().requires_grad_(),
          "f.layer1.1.bias": (rnd(C, seed=5) * 0.1).double().requires_grad_(),
          "f.layer2.0.weight": rnd(1, 2 + D, 1, 1, seed=6, scale=0.5).double().requires_grad_()}
    xr = x.double().requires_grad_()
    o = eb4.dynamic_filter(xr, diff.double(), sd, "f", True, k // 2)
    g_out, g_mask = rnd(*o["out"].shape, seed=7), rnd(*o["mask"].shape, seed=8)
    ((o["out"] * g_out.double()).sum() + (o["mask"] * g_mask.double()).sum()).backward()
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    diff_p = to_pix(diff).to(dev)

    def fn(t, a, w1, g1, b1, w2):
        if k == 1:
            proj = T.conv1x1(t, a, w1)
        else:
            proj = T.conv_dense(t, a, w1, 1, 1, 1, h, w_)
        proj = T.batchnorm_act(t, proj, g1, b1, rm, rv, 1e-5, 0.1, True, 1)
        return T.dynamic_filter(t, a, proj, diff_p, w2)
    params = [sd[k_].detach().float().to(dev).requires_grad_() for k_ in
              ("f.layer1.0.weight", "f.layer1.1.weight", "f.layer1.1.bias", "f.layer2.0.weight")]
    outs, gin, gp = run_tape(fn, [to_pix(x).to(dev)], params,
                             lambda o_: [to_pix(g_out).to(dev), to_pix(g_mask).to(dev)])
    check("out", to_nchw(outs[0]), o["out"])
    check("mask", to_nchw(outs[1]), o["mask"])
    check("dx", to_nchw(gin[0]), xr.grad)
    for p_, k_ in zip(gp, ("f.layer1.0.weight", "f.layer1.1.weight", "f.layer1.1.bias", "f.layer2.0.weight")):
        check("d " + k_, p_, sd[k_].grad)


# ---------------------------------------------------------------------------------------------
# large real 2-D FFT of image planes (csrc/fft_large.hip): loss tail + amplitude transfer at 128 / 256 / 320
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("S", [128, 256, 320])
@pytest.mark.parametrize("P", [1, 7])
def test_rfft2_planes_large(S, P):
    """ud_rfft2_planes vs torch.fft.rfft2 (float64, norm='ortho') and its adjoint vs the transpose identity
    <Y(x), G> == <x, adj(G)> and vs autograd; <= 1e-5 of the largest entry."""
    from unidefense_amd import kernels as K
    dev = _dev()
    g = torch.Generator().manual_seed(S + P)
    x = torch.randn(P, S, S, generator=g)
    Y = K.dft_rfft2_planes(x.to(dev).contiguous())
    Wh, Whp = S // 2 + 1, -(-(S // 2 + 1) // 4) * 4
    assert tuple(Y.shape) == (P, 2 * S, Whp)
    ref = torch.fft.rfft2(x.double(), norm="ortho")                       # [P, S, Wh]
    got_re, got_im = Y[:, :S, :Wh].double().cpu(), Y[:, S:, :Wh].double().cpu()
    scale = float(ref.abs().max())
    assert float((got_re - ref.real).abs().max()) <= 1e-5 * scale
    assert float((got_im - ref.imag).abs().max()) <= 1e-5 * scale
    assert float(Y[:, :, Wh:].abs().max()) == 0.0                          # padding columns
    # adjoint: gradient of sum(G * Y) w.r.t. x through torch's rfft2
    G = torch.randn(P, 2 * S, Whp, generator=g)
    G[:, :, Wh:] = 0
    xd = x.double().requires_grad_(True)
    Yd = torch.fft.rfft2(xd, norm="ortho")
    (Yd.real * G[:, :S, :Wh].double()).sum().add((Yd.imag * G[:, S:, :Wh].double()).sum()).backward()
    adj = K.dft_rfft2_planes_adjoint(G.to(dev).contiguous(), S).double().cpu()
    assert float((adj - xd.grad).abs().max()) <= 1e-5 * float(xd.grad.abs().max())


@pytest.mark.parametrize("N,H,W,Ho,Wo,Cc", [(2, 95, 95, 48, 48, 8), (1, 14, 9, 5, 4, 12), (3, 7, 7, 7, 7, 4), (2, 24, 24, 12, 12, 16),
                                            (1, 95, 95, 1, 1, 4)])
def test_adaptive_avgpool_any_size(N, H, W, Ho, Wo, Cc):
    """csrc/pool.hip: F.adaptive_avg_pool2d to a size that does not divide the input (model/efficientnet/exp.py:61-62 at 380 x 380:
    the 95 x 95 map pooled to 48 x 48 with overlapping windows of 2 and 3) and its adjoint, against torch in float64."""
    from unidefense_amd import kernels as K
    dev = _dev()
    g = torch.Generator().manual_seed(H * 100 + Ho)
    x = torch.randn(N, H, W, Cc, generator=g)
    dy = torch.randn(N, Ho, Wo, Cc, generator=g)
    xd = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    ref = torch.nn.functional.adaptive_avg_pool2d(xd, (Ho, Wo))
    (ref * dy.double().permute(0, 3, 1, 2)).sum().backward()
    y = K.adaptive_avgpool_fwd(x.to(dev), Ho, Wo)
    dx = K.adaptive_avgpool_bwd(dy.to(dev), H, W)
    e1 = (y.double().cpu() - ref.detach().permute(0, 2, 3, 1)).abs().max().item()
    e2 = (dx.double().cpu() - xd.grad.permute(0, 2, 3, 1)).abs().max().item()
    assert e1 < 1e-6 and e2 < 1e-6, (e1, e2)


def test_weight_layout_batch_matches_permutes():
    """kernels._WeightLayoutBatch / ud_weight_layouts_multi: the [rows][tap][channel] matrices of all registered k x k conv weights
    from ONE launch at the start of a forward — bit-equal to torch's permute / flip + contiguous, re-made from the CURRENT weights by
    every begin_forward, handed out only inside the forward and only while the parameter is the one they were made from."""
    from unidefense_amd import kernels as K
    dev = _dev()
    torch.manual_seed(5)
    owner = torch.nn.Module()
    ws = [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in ((20, 12, 3, 3), (3, 20, 3, 3), (8, 8, 1, 1), (48, 3, 3, 3), (16, 24, 5, 5))]

    def ref(w, mode):
        A, B, KH, KW = w.shape
        if mode == 0:
            return w.permute(0, 2, 3, 1).reshape(A, KH * KW * B).contiguous()
        if mode == 1:
            return w.flip(2, 3).permute(1, 2, 3, 0).reshape(B, KH * KW * A).contiguous()
        return w.permute(1, 2, 3, 0).reshape(B, KH * KW * A).contiguous()
    K.begin_forward(owner)
    for w in ws:                                   # first forward: made on the spot, and registered
        for mode in (0, 1, 2):
            assert torch.equal(K.weight_layout(w, mode), ref(w.detach(), mode))
    K.end_forward()
    batch = owner.__dict__["_ud_weight_layouts"]
    assert len(batch.entries) == 15 and batch.dirty
    with torch.no_grad():
        for w in ws:
            w.mul_(1.5)                            # an optimizer step in between
    K.begin_forward(owner)
    assert batch.active and not batch.dirty
    for w in ws:
        for mode in (0, 1, 2):
            got = K.weight_layout(w, mode)
            assert got.data_ptr() == batch.entries[(id(w), mode)][1].data_ptr()          # the batch's buffer, not a fresh permute
            assert torch.equal(got, ref(w.detach(), mode))
    K.end_forward()
    w = ws[0]
    assert batch.get(w, 0) is None                 # outside a forward nothing is handed out
    K.begin_forward(owner)
    with torch.no_grad():
        w.add_(1.0)                                # changed after begin(): the stale buffer must not be used
    assert batch.get(w, 0) is None and torch.equal(K.weight_layout(w, 0), ref(w.detach(), 0))
    K.end_forward()


@pytest.mark.parametrize("N,H,Ci,Co,k,stride,pad", [(8, 10, 128, 96, 3, 1, 1), (4, 16, 128, 160, 3, 2, 1), (2, 8, 64, 64, 5, 1, 2),
                                                    (16, 10, 256, 128, 3, 1, 1)])
def test_conv_as_im2col_planes_gemm(N, H, Ci, Co, k, stride, pad):
    """k x k conv = 1x1 conv over its im2col matrix on the planes GEMM (round 5: kernels.im2col_planes -> ud_gemm_p3; data gradient =
    GEMM + ud_col2im) — the path tape.conv_dense_any takes for Cin % 32 == 0 and KH KW Cin >= 1152: output, data gradient and
    weight gradient against F.conv2d in float64, stride 1 and 2, 3 x 3 and 5 x 5; and the SAME conv on the in-kernel-split gather
    GEMM (the path it replaces) for the record."""
    dev = _dev()
    from unidefense_amd import kernels as Kk
    from unidefense_amd import tape as T
    x = rnd(N, Ci, H, H, seed=11)
    w = rnd(Co, Ci, k, k, seed=12, scale=0.05)
    Ho = (H + 2 * pad - k) // stride + 1
    gy = rnd(N, Co, Ho, Ho, seed=13)
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    yr = F.conv2d(xr, wr, None, stride, pad)
    yr.backward(gy.double())
    g = Kk.conv_geom(N, H, H, Ci, Ho, Ho, k, k, stride, pad, pad, 0)
    min_flop = Kk._CONV_IM2COL_MIN_FLOP
    for on in (True, False):
        Kk._CONV_IM2COL, Kk._CONV_IM2COL_MIN_FLOP = on, 0.0          # (the test shapes are below the launch-bound threshold)
        try:
            assert Kk.conv_im2col_ok(g, Co, to_pix(x).to(dev)) == on
            outs, gin, gp = run_tape(lambda t, a, b: T.conv_dense_any(t, a, b, stride, pad), [to_pix(x).to(dev)], [w.to(dev)],
                                     lambda o: [to_pix(gy).to(dev)])
        finally:
            Kk._CONV_IM2COL, Kk._CONV_IM2COL_MIN_FLOP = True, min_flop
        tag = "im2col planes" if on else "gather"
        check(f"{tag}: y", to_nchw(outs[0]), yr)
        check(f"{tag}: dx", to_nchw(gin[0]), xr.grad)
        check(f"{tag}: dw", gp[0], wr.grad)


@pytest.mark.parametrize("G,R,C,act", [(32, 16384, 20, 1), (32, 4096, 40, 1), (32, 1024, 80, 1), (32, 256, 80, 1),      # the decoder's InstanceNorms
                                         (1, 8 * 64 * 64, 64, 2), (1, 16 * 40 * 40, 512, 2), (1, 2048, 272, 1),           # BatchNorm shapes
                                         (3, 1000, 12, 0), (1, 37, 4, 1), (5, 7, 260, 2), (1, 100000, 8, 0)])            # ragged
def test_one_launch_norm_equals_three_launch_form(G, R, C, act):
    """csrc/norm.hip norm_fwd_fused / norm_bwd_fused (statistics + apply in one kernel, the workgroups of a column group exchanging
    their partial sums through agent-scope atomics) against float64 and against the three-launch kernels they replace, twice in a
    row on the same scratch (the counters are fresh zeros per call) — y, mean, invstd, running statistics, dx, dgamma, dbeta."""
    dev = _dev()
    from tests.margins import within
    from unidefense_amd import kernels as K
    x = (rnd(G * R, C, seed=1) * 1.5 + 0.3).to(dev)
    dy = rnd(G * R, C, seed=2).to(dev)
    ga, be = (rnd(C, seed=3) * 0.1 + 1).to(dev), (rnd(C, seed=4) * 0.1).to(dev)
    assert K.norm_fused_takes(x, G, R)
    for rep in range(2):
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        rm0, rv0 = rm.clone(), rv.clone()
        y, mean, invstd = K.norm_fwd_fused(x, G, R, ga, be, act, 1e-5, 0.1, rm if G == 1 else None, rv if G == 1 else None)
        mean3, invstd3 = K.norm_stats(x, G, R, 1e-5, 0.1, rm0 if G == 1 else None, rv0 if G == 1 else None)
        y3 = K.norm_apply(x, G, R, mean3, invstd3, ga, be, act)
        dx, dg, db = K.norm_bwd_fused(x, dy, G, R, mean, invstd, ga, be, act)
        dx3, dg3, db3 = K.norm_bwd(x, dy, G, R, mean3, invstd3, ga, be, act)
        torch.cuda.synchronize()
        # float64 reference
        xd = x.double().view(G, R, C).requires_grad_()
        gd, bd = ga.double().requires_grad_(), be.double().requires_grad_()
        mu = xd.mean(1, keepdim=True)
        var = xd.var(1, unbiased=False, keepdim=True)
        z = (xd - mu) / torch.sqrt(var + 1e-5) * gd + bd
        yd = z * torch.sigmoid(z) if act == 1 else (torch.relu(z) if act == 2 else z)
        yd.backward(dy.double().view(G, R, C))
        for name, got, got3, want in (("y", y, y3, yd.view(G * R, C)), ("mean", mean, mean3, mu.view(G, C)),
                                      ("invstd", invstd, invstd3, (1 / torch.sqrt(var + 1e-5)).view(G, C)),
                                      ("dx", dx, dx3, xd.grad.view(G * R, C)), ("dgamma", dg, dg3, gd.grad), ("dbeta", db, db3, bd.grad)):
            sc = float(want.detach().abs().max().clamp_min(1e-6))
            e = float((got.double() - want.detach()).abs().max()) / sc
            e3 = float((got3.double() - want.detach()).abs().max()) / sc
            assert within(f"fused norm {name}", e, max(2e-5, 3 * e3)), (name, e, e3)
        if G == 1:
            assert within("fused norm running_mean", float((rm - rm0).abs().max()), 1e-6)
            assert within("fused norm running_var", float((rv - rv0).abs().max() / rv0.abs().max()), 1e-6)
