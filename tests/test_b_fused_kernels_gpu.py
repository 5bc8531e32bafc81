"""GPU, operator level: every entry point of the fused MBConv path (csrc/fused.hip, the fused FFT variants of csrc/fft.hip,
the GEMM epilogue statistics, the half-storage instantiations) through the C ABI against torch in float64 — at ragged shapes:
channel counts that do not fill a column group, row counts that are not multiples of the chunking, one and several samples,
both forms of every reduction (fp64 atomics / partials + finalize).  Reference semantics: model/efficientnet/model.py:94-135
(MBConvBlock.forward), exp.py:46-65 (SFConv), utils.py:66-77 (swish)."""
import pytest
from tests.margins import within
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


# ---------------------------------------------------------------------------------------------
# kernel-level parity of the deferred-normalisation entry points (csrc/fused.hip, through the C ABI) against torch in
# float64, at ragged shapes: channel counts that do not fill a column group, row counts that are not multiples of the
# chunking, one and several samples, both forms of every reduction (fp64 atomics / partials + finalize).
# ---------------------------------------------------------------------------------------------
SHAPES = [(1, 37, 8), (3, 50, 24), (2, 257, 40), (4, 1024, 144), (2, 4096, 272), (32, 64, 1632), (2, 70000, 48)]


def _mk(G, R, Cc, seed, dev):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(G, R, Cc, generator=g) * 1.7 + 0.3).to(dev)
    gamma = (1.0 + 0.2 * torch.randn(Cc, generator=g)).to(dev)
    beta = (0.1 * torch.randn(Cc, generator=g)).to(dev)
    return g, x, gamma, beta


def _bn_ref(x, gamma, beta, eps, act):
    xd = x.double()
    mean = xd.mean((0, 1))
    var = xd.var((0, 1), unbiased=False)
    z = (xd - mean) / torch.sqrt(var + eps) * gamma.double() + beta.double()
    return z * torch.sigmoid(z) if act else z


def _deferred(K, x, gamma, beta, eps, act, rm=None, rv=None, mom=0.0):
    G, R, Cc = x.shape
    acc = K.zeros64(2 * Cc, x)
    K.colstats(x.view(G * R, Cc), acc)
    return K.DeferredBN(acc, Cc, G * R, gamma, beta, eps, act, mom, rm, rv)


@pytest.mark.parametrize("G,R,Cc", SHAPES)
def test_deferred_bn_forward_kernels(G, R, Cc):
    from unidefense_amd import kernels as K
    dev = _dev()
    K.reset_zero_pool()
    g, x, gamma, beta = _mk(G, R, Cc, 11 + Cc, dev)
    eps = 1e-3
    rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    bn = _deferred(K, x, gamma, beta, eps, 1, rm, rv, 0.01)
    ref = _bn_ref(x, gamma, beta, eps, True)
    # statistics
    xd = x.double()
    assert _rel(bn.acc[:Cc], xd.sum((0, 1))) < 1e-12 and _rel(bn.acc[Cc:], (xd * xd).sum((0, 1))) < 1e-12
    # materialised apply (also moves the running statistics, once)
    y = K.bn_apply(x, bn, G, R, update=True)
    assert _rel(y, ref) < 5e-6
    n = G * R
    assert _rel(rm, 0.01 * xd.mean((0, 1))) < 1e-5
    assert _rel(rv, 0.99 + 0.01 * xd.var((0, 1), unbiased=False) * n / max(n - 1, 1)) < 1e-5
    # SE pooling sums, gate and gated activation
    pool = K.zeros64(G * Cc, x)
    K.colsum_bn(x, bn, G, R, pool)
    assert _rel(pool.view(G, Cc), ref.sum(1)) < 5e-6
    s = torch.randn(G, Cc, generator=g).to(dev)
    yg = K.se_scale_bn(x, bn, s, G, R)
    assert _rel(yg, ref * torch.sigmoid(s.double())[:, None, :]) < 5e-6
    dy = torch.randn(G, R, Cc, generator=g).to(dev)
    dot = K.zeros64(G * Cc, x)
    K.coldot_bn(dy, x, bn, G, R, dot)
    assert _rel(dot.view(G, Cc), (dy.double() * ref).sum(1)) < 5e-6
    # BN2 + drop-connect + skip
    bn0 = _deferred(K, x, gamma, beta, eps, 0)
    keep = (torch.rand(G, generator=g) < 0.7).float().to(dev)
    skip = torch.randn(G, R, Cc, generator=g).to(dev)
    out = K.residual_bn(x, bn0, keep, 1.25, skip, G, R)
    want = _bn_ref(x, gamma, beta, eps, False) * (keep.double() * 1.25)[:, None, None] + skip.double()
    assert _rel(out, want) < 5e-6


@pytest.mark.parametrize("G,R,Cc", SHAPES)
@pytest.mark.parametrize("act", [0, 1])
def test_deferred_bn_backward_kernels(G, R, Cc, act):
    """normbwd_sums / normbwd_apply (with the drop-connect scale) and the fused se_scale_bwd_bn against autograd."""
    from unidefense_amd import kernels as K
    dev = _dev()
    K.reset_zero_pool()
    g, x, gamma, beta = _mk(G, R, Cc, 23 + Cc + act, dev)
    eps = 1e-3
    dy = torch.randn(G, R, Cc, generator=g).to(dev)
    keep = (torch.rand(G, generator=g) < 0.7).float().to(dev)
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    mean, var = xd.mean((0, 1)), xd.var((0, 1), unbiased=False)
    z = (xd - mean) / torch.sqrt(var + eps) * gd + bd
    y = z * torch.sigmoid(z) if act else z
    (y * (keep.double() * 1.25)[:, None, None] * dy.double()).sum().backward()
    bn = _deferred(K, x, gamma, beta, eps, act)
    sacc = K.zeros64(2 * Cc, x)
    K.normbwd_sums(x, dy, keep, 1.25, bn, False, G, R, sacc)
    dx, dg, db = K.normbwd_apply(x, dy, keep, 1.25, bn, False, G, R, sacc)
    assert _rel(dx, xd.grad) < 2e-5 and _rel(dg, gd.grad) < 2e-5 and _rel(db, bd.grad) < 2e-5
    if act:
        # gate path: y = swish(bn(x)) * sigmoid(s);  upstream dc;  pooled branch dpool / HW
        s = torch.randn(G, Cc, generator=g).to(dev)
        dpool = torch.randn(G, Cc, generator=g).to(dev)
        dc = torch.randn(G, R, Cc, generator=g).to(dev)
        xd2 = x.double().requires_grad_(True)
        m2, v2 = xd2.mean((0, 1)), xd2.var((0, 1), unbiased=False)
        z2 = (xd2 - m2) / torch.sqrt(v2 + eps) * gamma.double() + beta.double()
        b = z2 * torch.sigmoid(z2)
        ((b * torch.sigmoid(s.double())[:, None, :] * dc.double()).sum() + (b.mean(1) * dpool.double()).sum()).backward()
        sacc2 = K.zeros64(2 * Cc, x)
        dz = K.se_scale_bwd_bn(dc, x, bn, s, dpool, 1.0 / R, G, R, sacc2)
        dx2, _, _ = K.normbwd_apply(x, dz, None, 1.0, bn, True, G, R, sacc2)
        assert _rel(dx2, xd2.grad) < 2e-5


@pytest.mark.parametrize("N,Cc,Cs", [(4, 24, 6), (32, 1632, 68), (5, 448, 112), (16, 192, 8)])
def test_se_backward_kernels(N, Cc, Cs):
    """ud_se_bwd_a / ud_se_bwd_b: the two FC layers of the squeeze-excite backward against autograd (float64)."""
    from unidefense_amd import kernels as K
    dev = _dev()
    K.reset_zero_pool()
    g = torch.Generator().manual_seed(N + Cc)
    HW = 49
    pool_sum = (torch.randn(N, Cc, generator=g) * HW).double()
    Wr = (torch.randn(Cs, Cc, generator=g) / Cc ** 0.5)
    br = torch.randn(Cs, generator=g) * 0.1
    We = (torch.randn(Cc, Cs, generator=g) / Cs ** 0.5)
    be = torch.randn(Cc, generator=g) * 0.1
    dgate = torch.randn(N, Cc, generator=g).double()
    Wrd, brd, Wed, bed = (t.double().requires_grad_(True) for t in (Wr, br, We, be))
    pool = (pool_sum / HW).requires_grad_(True)
    s1 = pool @ Wrd.t() + brd
    s2 = (s1 * torch.sigmoid(s1)) @ Wed.t() + bed
    (torch.sigmoid(s2) * dgate).sum().backward()
    dpool, dWe, dbe, dWr, dbr = K.se_bwd(dgate.to(dev), s2.detach().float().to(dev), s1.detach().float().to(dev),
                                         We.to(dev), Wr.to(dev), pool_sum.to(dev), 1.0 / HW)
    for got, want, name in ((dpool, pool.grad, "dpool"), (dWe, Wed.grad, "dWe"), (dbe, bed.grad, "dbe"),
                            (dWr, Wrd.grad, "dWr"), (dbr, brd.grad, "dbr")):
        assert _rel(got, want) < 2e-5, (name, _rel(got, want))
    # forward FCs of the same block
    s1g = K.fc_fwd_d(pool_sum.to(dev), 1.0 / HW, Wr.to(dev), br.to(dev), N)
    assert _rel(s1g, s1) < 1e-5
    s2g = K.fc_fwd(s1g, We.to(dev), be.to(dev), 1)
    assert _rel(s2g, s2) < 1e-5


@pytest.mark.parametrize("N,H,Cc,k,stride,pad", [(2, 16, 24, 3, 1, (1, 1, 1, 1)), (2, 16, 40, 5, 1, (2, 2, 2, 2)),
                                                 (1, 32, 48, 3, 2, (0, 1, 0, 1)), (2, 16, 144, 5, 2, (1, 2, 1, 2)),
                                                 (3, 9, 8, 3, 1, (1, 1, 1, 1))])
@pytest.mark.parametrize("gate_mode", [0, 2])
def test_dwconv_backward_through_deferred_bn(N, H, Cc, k, stride, pad, gate_mode):
    """ud_dwconv_bwd_data_bn (gate * conv^T(dy) + add, pushed through swish'(bn(x)), BN sums) and ud_dwconv_bwd_weight_ex
    against autograd; ud_dwconv_bwd_data_ex likewise."""
    import torch.nn.functional as F
    from unidefense_amd import kernels as K
    dev = _dev()
    K.reset_zero_pool()
    g = torch.Generator().manual_seed(N * 100 + Cc + k)
    pl, pr, pt, pb = pad
    x = torch.randn(N, H, H, Cc, generator=g)
    w = torch.randn(Cc, 1, k, k, generator=g) * 0.3
    gamma, beta = 1.0 + 0.2 * torch.randn(Cc, generator=g), 0.1 * torch.randn(Cc, generator=g)
    alpha = torch.tensor(0.4)
    Ho = (H + pt + pb - k) // stride + 1
    dy = torch.randn(N, Ho, Ho, Cc, generator=g)
    add = torch.randn(N, H, H, Cc, generator=g)
    gate = float(1 - torch.sigmoid(alpha)) if gate_mode == 2 else 1.0
    xd = x.double().requires_grad_(True)
    mean, var = xd.mean((0, 1, 2)), xd.var((0, 1, 2), unbiased=False)
    z = (xd - mean) / torch.sqrt(var + 1e-3) * gamma.double() + beta.double()
    a = (z * torch.sigmoid(z)).detach().requires_grad_(True)           # stop at the activated tensor: da is what we want
    wd = w.double().requires_grad_(True)
    yc = F.conv2d(F.pad(a.permute(0, 3, 1, 2), (pl, pr, pt, pb)), wd, stride=stride, groups=Cc).permute(0, 2, 3, 1)
    ((yc * dy.double()).sum() * gate).backward()
    da = a.grad + add.double()
    zc = z.detach()
    sg = torch.sigmoid(zc)
    dz_ref = da * (sg * (1 + zc * (1 - sg)))
    xh = (x.double() - mean.detach()) / torch.sqrt(var.detach() + 1e-3)
    xg, dyg, addg = x.to(dev), dy.to(dev), add.to(dev)
    wt = w.view(Cc, k * k).t().contiguous().to(dev)
    bn = _deferred(K, xg.view(N, H * H, Cc), gamma.to(dev), beta.to(dev), 1e-3, 1)
    al = alpha.to(dev)
    if stride == 1:
        sacc = K.zeros64(2 * Cc, xg)
        dz = K.dwconv_bwd_data_bn(dyg, al if gate_mode else None, gate_mode, wt, addg, xg, bn, k, stride, pt, pl, sacc)
        assert _rel(dz, dz_ref) < 2e-5
        assert _rel(sacc[:Cc], dz_ref.sum((0, 1, 2))) < 2e-5 and _rel(sacc[Cc:], (dz_ref * xh).sum((0, 1, 2))) < 2e-5
    dxe = K.dwconv_bwd_data_ex(dyg, al if gate_mode else None, gate_mode, wt, addg, k, stride, pt, pl, H, H)
    assert _rel(dxe, da) < 2e-5
    ag = a.detach().float().to(dev)
    dw = K.dwconv_bwd_weight_ex(ag, dyg, al if gate_mode else None, gate_mode, k, stride, pt, pl)
    assert _rel(dw.view(Cc, k, k), wd.grad.view(Cc, k, k)) < 2e-5


@pytest.mark.parametrize("S,Cc,N", [(8, 24, 3), (16, 40, 2), (32, 48, 2), (64, 8, 1), (12, 48, 3), (24, 44, 2), (48, 24, 2), (20, 24, 2)])
def test_fft_fused_variants(S, Cc, N):
    """ud_rfft2_ex (deferred BN + swish on load, activated side output, gate factor) and ud_irfft2_mix (irfft2 + SF mix +
    BN statistics) against torch.fft in float64."""
    from unidefense_amd import kernels as K
    dev = _dev()
    K.reset_zero_pool()
    g = torch.Generator().manual_seed(S + Cc)
    x = torch.randn(N, S, S, Cc, generator=g).to(dev)
    gamma, beta = (1.0 + 0.2 * torch.randn(Cc, generator=g)).to(dev), (0.1 * torch.randn(Cc, generator=g)).to(dev)
    alpha = torch.tensor(-0.3, device=dev)
    bn = _deferred(K, x.view(N, S * S, Cc), gamma, beta, 1e-3, 1)
    act_ref = _bn_ref(x.view(N, S * S, Cc), gamma, beta, 1e-3, True).view(N, S, S, Cc)
    Y, act = K.rfft2_ex(x, 1.0 / S, 1.0, bn=bn, want_act=True, gate_alpha=alpha, gate_mode=1)
    assert _rel(act, act_ref) < 5e-6
    F_ = torch.fft.rfft2(act_ref.permute(0, 3, 1, 2), norm="ortho").permute(0, 2, 3, 1) * torch.sigmoid(alpha.double().cpu())
    assert _rel(Y[..., :Cc], F_.real) < 2e-5 and _rel(Y[..., Cc:], F_.imag) < 2e-5
    # irfft2 + mix + statistics
    Yin = torch.randn(N, S, S // 2 + 1, 2 * Cc, generator=g).to(dev)
    spat = torch.randn(N, S, S, Cc, generator=g).to(dev)
    acc = K.zeros64(2 * Cc, x)
    y, fr = K.irfft2_mix(Yin, 1.0 / S, spat, alpha, acc)
    Yc = torch.complex(Yin[..., :Cc].double().cpu(), Yin[..., Cc:].double().cpu()).permute(0, 3, 1, 2)
    fr_ref = torch.fft.irfft2(Yc, s=(S, S), norm="ortho").permute(0, 2, 3, 1)
    a = torch.sigmoid(alpha.double().cpu())
    y_ref = (1 - a) * spat.double().cpu() + a * fr_ref
    assert _rel(fr, fr_ref - spat.double().cpu()) < 2e-5 and _rel(y, y_ref) < 2e-5          # second output: freq - spat
    assert _rel(acc[:Cc], y_ref.sum((0, 1, 2))) < 2e-5 and _rel(acc[Cc:], (y_ref * y_ref).sum((0, 1, 2))) < 2e-5


@pytest.mark.parametrize("half", [False, True], ids=["fp32", "half"])
@pytest.mark.parametrize("S,Cc,N", [(64, 144, 3), (64, 200, 2), (32, 192, 3), (32, 40, 2)])
def test_fft_two_pass_equals_one_kernel(S, Cc, N, half):
    """The two-pass forms (row kernel + column kernel, half-spectrum in HBM: ud_rfft2_two_pass / ud_irfft2_two_pass) run the
    same butterflies on the same values as the LDS-resident kernels: every tensor they write equals the one-kernel form's
    to the LAST BIT OR TWO (hipcc contracts the scale factor into the first butterfly in one form and not in the other: 1 ulp,
    observed on ~1e-6 of the elements), plain and with every fused prologue / epilogue (deferred BN + swish on load, activated
    copy, running statistics, gate factor, gate gradient from the slots; SF mix + freq - spat + BN sums), channel counts that
    are not a multiple of the 64 a wave owns, both storage types.  Against float64: test_fft_fused_variants (whichever form
    the policy picks) and test_half_storage_kernels."""
    from unidefense_amd import kernels as K
    dev = _dev()
    K.reset_zero_pool()
    dt = torch.float16 if half else torch.float32
    g = torch.Generator().manual_seed(S + Cc)
    x = torch.randn(N, S, S, Cc, generator=g).to(dev).to(dt)
    gamma, beta = (1.0 + 0.2 * torch.randn(Cc, generator=g)).to(dev), (0.1 * torch.randn(Cc, generator=g)).to(dev)
    alpha = torch.tensor(-0.3, device=dev)
    Yin = torch.randn(N, S, S // 2 + 1, 2 * Cc, generator=g).to(dev).to(dt)
    spat = torch.randn(N, S, S, Cc, generator=g).to(dev).to(dt)
    slots = torch.randn(64, generator=g, dtype=torch.float64).to(dev)

    def run(two_pass):
        saved = K._FFT_TWO_PASS
        K._FFT_TWO_PASS = two_pass
        try:
            out = {}
            out["rfft"] = K.rfft2(x, 1.0 / S, 2.0)
            out["irfft"] = K.irfft2(Yin, 1.0 / S, 0.5)
            rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
            bn = _deferred(K, x.view(N, S * S, Cc), gamma, beta, 1e-3, 1, rm, rv, 0.01)
            out["ex Y"], out["ex act"] = K.rfft2_ex(x, 1.0 / S, 1.0, bn=bn, want_act=True, update=True)
            out["ex running_mean"], out["ex running_var"] = rm, rv
            out["gate Y"], _, out["gate grad"] = K.rfft2_ex(x, 1.0 / S, 2.0, gate_alpha=alpha, gate_mode=1, gate_acc=slots)
            out["gate2 Y"], _ = K.rfft2_ex(x, 1.0 / S, 1.0, gate_alpha=alpha, gate_mode=2)
            acc = K.zeros64(2 * Cc, x)
            out["mix y"], out["mix diff"] = K.irfft2_mix(Yin, 1.0 / S, spat, alpha, acc)
            torch.cuda.synchronize()
            return out, acc.clone()
        finally:
            K._FFT_TWO_PASS = saved
    o1, a1 = run(False)
    o2, a2 = run(True)
    ulp = 2.0 ** -10 if half else 2.0 ** -23
    worst = 0.0
    for k in o1:
        scale = float(o1[k].double().abs().max())
        e = float((o1[k].double() - o2[k].double()).abs().max()) / (scale if scale > 0 else 1.0)
        worst = max(worst, e / ulp)
        assert e <= 2.0 * ulp, (k, e)
    within(f"two-pass vs one-kernel, worst |difference| / max|value| in ulps of the storage type", worst, 2.0)
    assert _rel(a2, a1) < (1e-3 if half else 1e-6)          # sums of y and y^2 whose last bits differ as above


@pytest.mark.parametrize("M,N,Kd", [(2048, 1632, 272), (8192 + 40, 960, 160), (131072, 192, 32), (524288 // 4, 144, 24),
                                   (300, 48, 24), (2048, 272, 1632)])
def test_gemm_epilogue_statistics(M, N, Kd):
    """ud_gemm with stat_sum / stat_sumsq: the BatchNorm sums of the GEMM's result out of its epilogue (direct adds up to
    64 row tiles, 64 slots + fold beyond; split plans fall back to ud_colstats) against the sums of the stored result."""
    from unidefense_amd import kernels as K
    dev = _dev()
    K.reset_zero_pool()
    g = torch.Generator().manual_seed(M % 1000 + N)
    a = torch.randn(M, Kd, generator=g).to(dev)
    w = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(dev)
    acc = K.zeros64(2 * N, a)
    y, done = K.gemm_nt(a, w, stats=acc)
    if not done:
        K.colstats(y, acc)
    yd = y.double()
    assert _rel(y, a.double() @ w.double().t()) < 2e-5
    assert _rel(acc[:N], yd.sum(0)) < 1e-9 and _rel(acc[N:], (yd * yd).sum(0)) < 1e-9
    print(f"  {M}x{N}x{Kd}: epilogue statistics {'used' if done else 'not applicable (split plan)'}")


# ------------------------------------------------------------------------------------------------ half storage
def _bn_for(Cc, acc, count, act, dev, seed=0):
    from unidefense_amd import kernels as K
    g = torch.Generator().manual_seed(seed)
    gamma = (0.5 + torch.rand(Cc, generator=g)).to(dev)
    beta = (0.2 * torch.randn(Cc, generator=g)).to(dev)
    return K.DeferredBN(acc, Cc, count, gamma, beta, 1e-3, act)


@pytest.mark.parametrize("shape", [(4, 16, 16, 96), (2, 32, 32, 48), (3, 8, 8, 272)])
def test_half_storage_kernels_match_fp32_on_rounded_inputs(shape):
    """Every kernel of the fused MBConv path instantiated for _Float16 storage against its fp32 instantiation fed the
    SAME (fp16-representable) inputs: the arithmetic is identical (fp32 registers), so results agree to one rounding of the
    stored value (2^-11 relative) and the fp64 statistics to ~1e-3 of their scale (they are taken of the rounded values)."""
    from unidefense_amd import kernels as K
    dev = _dev()
    K.reset_zero_pool()
    N, H, W, Cc = shape
    M, HW = N * H * W, H * W
    g = torch.Generator().manual_seed(Cc + H)

    def rnd(*s):
        return torch.randn(*s, generator=g).half().to(dev)          # fp16-representable values
    x16, dy16 = rnd(N, H, W, Cc), rnd(N, H, W, Cc)
    x32, dy32 = x16.float(), dy16.float()

    def close(a, b, tol=2e-3, what=""):
        e = _rel(a.double(), b.double())
        assert e < tol, (what, e)

    # statistics
    a16, a32 = K.zeros64(2 * Cc, x16), K.zeros64(2 * Cc, x16)
    K.colstats(x16.view(M, Cc), a16)
    K.colstats(x32.view(M, Cc), a32)
    close(a16, a32, 1e-12, "colstats")
    bn16, bn32 = _bn_for(Cc, a16, M, 1, dev), _bn_for(Cc, a32, M, 1, dev)
    # elementwise consumers
    close(K.bn_apply(x16, bn16, 1, M), K.bn_apply(x32, bn32, 1, M), what="bn_apply")
    s = torch.randn(N, Cc, generator=g).to(dev)
    close(K.se_scale_bn(x16, bn16, s, N, HW), K.se_scale_bn(x32, bn32, s, N, HW), what="se_scale_bn")
    keep = (torch.rand(N, generator=g) < 0.8).float().to(dev)
    close(K.residual_bn(x16, bn16, keep, 1.25, dy16, N, HW), K.residual_bn(x32, bn32, keep, 1.25, dy32, N, HW),
          what="residual_bn")
    # reductions behind a deferred BatchNorm
    p16, p32 = K.zeros64(N * Cc, x16), K.zeros64(N * Cc, x16)
    K.colsum_bn(x16, bn16, N, HW, p16)
    K.colsum_bn(x32, bn32, N, HW, p32)
    close(p16, p32, 1e-9, "colsum_bn")
    d16, d32 = K.zeros64(N * Cc, x16), K.zeros64(N * Cc, x16)
    K.coldot_bn(dy16, x16, bn16, N, HW, d16)
    K.coldot_bn(dy32, x32, bn32, N, HW, d32)
    close(d16, d32, 1e-9, "coldot_bn")
    # BatchNorm backward
    sb16, sb32 = K.zeros64(2 * Cc, x16), K.zeros64(2 * Cc, x16)
    K.normbwd_sums(x16, dy16, keep, 1.25, bn16, False, N, HW, sb16)
    K.normbwd_sums(x32, dy32, keep, 1.25, bn32, False, N, HW, sb32)
    close(sb16, sb32, 1e-9, "normbwd_sums")
    r16 = K.normbwd_apply(x16, dy16, keep, 1.25, bn16, False, N, HW, sb16)
    r32 = K.normbwd_apply(x32, dy32, keep, 1.25, bn32, False, N, HW, sb32)
    close(r16[0], r32[0], what="normbwd_apply dx")
    close(r16[1], r32[1], 1e-6, "dgamma")
    dpool = (0.1 * torch.randn(N, Cc, generator=g)).to(dev)
    z16, z32 = K.zeros64(2 * Cc, x16), K.zeros64(2 * Cc, x16)
    dz16 = K.se_scale_bwd_bn(dy16, x16, bn16, s, dpool, 1.0 / HW, N, HW, z16)
    dz32 = K.se_scale_bwd_bn(dy32, x32, bn32, s, dpool, 1.0 / HW, N, HW, z32)
    close(dz16, dz32, what="se_scale_bwd_bn dz")
    close(z16, z32, 3e-3, "se_scale_bwd_bn sums")          # sums of the ROUNDED gradient vs of the unrounded one
    # depthwise conv, forward / data gradient (+ BatchNorm sums) / weight gradient
    for k in (3, 5):
        wt = (torch.randn(k * k, Cc, generator=g) / k).to(dev)
        pad = (k - 1) // 2
        y16, y32 = K.dwconv_fwd(x16, wt, k, 1, pad, pad, H, W), K.dwconv_fwd(x32, wt, k, 1, pad, pad, H, W)
        close(y16, y32, what=f"dwconv_fwd k{k}")
        q16, q32 = K.zeros64(2 * Cc, x16), K.zeros64(2 * Cc, x16)
        alpha = torch.tensor([0.3], device=dev)
        e16 = K.dwconv_bwd_data_bn(dy16, alpha, 2, wt, x16, x16, bn16, k, 1, pad, pad, q16)
        e32 = K.dwconv_bwd_data_bn(dy32, alpha, 2, wt, x32, x32, bn32, k, 1, pad, pad, q32)
        close(e16, e32, what=f"dwconv_bwd_data_bn k{k}")
        close(q16, q32, 3e-3, "its sums")
        close(K.dwconv_bwd_data_ex(dy16, alpha, 1, wt, None, k, 1, pad, pad, H, W),
              K.dwconv_bwd_data_ex(dy32, alpha, 1, wt, None, k, 1, pad, pad, H, W), what="dwconv_bwd_data_ex")
        close(K.dwconv_bwd_weight_ex(x16, dy16, alpha, 1, k, 1, pad, pad),
              K.dwconv_bwd_weight_ex(x32, dy32, alpha, 1, k, 1, pad, pad), 1e-5, "dwconv_bwd_weight_ex")
        close(K.dwconv_bwd_data(dy16[:, ::2, ::2].contiguous(), wt, k, 2, pad, pad, H, W, add=x16),
              K.dwconv_bwd_data(dy32[:, ::2, ::2].contiguous(), wt, k, 2, pad, pad, H, W, add=x32), what="dwconv_bwd_data s2")
    # FFTs (power-of-two maps)
    if H in (8, 16, 32, 64):
        Y16, act16 = K.rfft2_ex(x16, 1.0 / H, 1.0, bn=bn16, want_act=True)
        Y32, act32 = K.rfft2_ex(x32, 1.0 / H, 1.0, bn=bn32, want_act=True)
        close(act16, act32, what="rfft2_ex act")
        close(Y16, Y32, 3e-3, "rfft2_ex Y")                 # transform of the rounded activation vs of the unrounded one
        Yr = Y16.float()
        alpha = torch.tensor([0.3], device=dev)
        m16, m32 = K.zeros64(2 * Cc, x16), K.zeros64(2 * Cc, x16)
        o16 = K.irfft2_mix(Y16, 1.0 / H, x16, alpha, m16)
        o32 = K.irfft2_mix(Yr, 1.0 / H, x32, alpha, m32)
        close(o16[0], o32[0], what="irfft2_mix y")
        close(o16[1], o32[1], what="irfft2_mix freq")
        close(m16, m32, 3e-3, "irfft2_mix sums")
        close(K.irfft2(Y16, 1.0 / H, 0.5), K.irfft2(Yr, 1.0 / H, 0.5), what="irfft2")
        close(K.rfft2(x16, 1.0 / H, 2.0), K.rfft2(x32, 1.0 / H, 2.0), what="rfft2")
        sp16 = K.sfmix_fwd(x16, dy16, alpha, False)
        close(sp16, K.sfmix_fwd(x32, dy32, alpha, False), what="sfmix_fwd")
        b16, b32 = K.sfmix_bwd(x16, dy16, alpha, x16, False), K.sfmix_bwd(x32, dy32, alpha, x32, False)
        close(b16[0], b32[0], what="sfmix_bwd dspat")
        close(b16[2], b32[2], 1e-5, "sfmix_bwd dalpha")
    close(K.axpby(x16, 0.5, dy16, 2.0), K.axpby(x32, 0.5, dy32, 2.0), what="axpby")


@pytest.mark.parametrize("M,N,Kd", [(2048, 1632, 272), (8192, 160, 960), (131072, 192, 32), (300, 48, 24), (2048, 272, 1632)])
def test_half_operand_gemms(M, N, Kd):
    """ud_gemm with half_mask: the three products of a 1x1 conv on half-stored activations (forward: A half, C half + the
    epilogue statistics; data gradient: A half, C half, plain and accumulating; weight gradient: A and B half, C fp32)
    against float64 on the same fp16-representable operands (fp16 MFMA products are exact, fp32 accumulation)."""
    from unidefense_amd import kernels as K
    dev = _dev()
    K.reset_zero_pool()
    g = torch.Generator().manual_seed(M % 977 + N)
    a = torch.randn(M, Kd, generator=g).half().to(dev)
    w = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).half().float().to(dev)       # fp32 weights holding fp16-exact values
    ref = a.double() @ w.double().t()
    acc = K.zeros64(2 * N, a)
    y, done = K.gemm_nt(a, w, stats=acc)
    assert y.dtype == torch.float16 and done
    assert _rel(y.double(), ref) < 1e-3                      # one fp16 rounding of the result
    yd = y.double()
    assert _rel(acc[:N], yd.sum(0)) < 1e-9 and _rel(acc[N:], (yd * yd).sum(0)) < 1e-9
    # data gradient dY @ W, plain and accumulating onto an existing gradient
    dy = torch.randn(M, N, generator=g).half().to(dev)
    refd = dy.double() @ w.double()
    dx = K.gemm_nn(dy, w)
    assert dx.dtype == torch.float16 and _rel(dx.double(), refd) < 1e-3
    base = torch.randn(M, Kd, generator=g).half().to(dev)
    dx2 = K.gemm_nn(dy, w, out=base.clone(), accumulate=True)
    assert _rel(dx2.double(), refd + base.double()) < 1e-3
    # weight gradient dY^T @ A: fp32 result
    dw = K.gemm_tn(dy, a)
    assert dw.dtype == torch.float32 and _rel(dw.double(), dy.double().t() @ a.double()) < 2e-6


# ------------------------------------------------------------------------------------------------ LDS-tiled depthwise
@pytest.mark.parametrize("N,H,W,Cc,k", [(2, 8, 8, 40, 5), (3, 8, 8, 24, 3), (2, 16, 16, 48, 5), (2, 16, 16, 36, 3),
                                        (1, 32, 32, 24, 5), (2, 64, 64, 8, 3), (2, 19, 23, 12, 5), (1, 9, 40, 68, 3),
                                        (1, 130, 17, 4, 3)])
@pytest.mark.parametrize("half", [False, True], ids=["fp32", "half"])
def test_tiled_depthwise_kernels(N, H, W, Cc, k, half):
    """csrc/dwtile.hip (stride 1, 'same' pads): forward with a deferred BatchNorm + swish on its input and the BN1
    statistics of its output; data gradient with gate, added gradient, act'(bn(x)) and the BatchNorm backward sums; weight
    gradient with the deferred BatchNorm on the staged input — against autograd in float64 at whole-image, multi-tile and
    ragged maps, both storage types (half: inputs are fp16-representable, results within one rounding)."""
    import torch.nn.functional as F
    from unidefense_amd import kernels as K
    dev = _dev()
    K.reset_zero_pool()
    g = torch.Generator().manual_seed(N * 1000 + H + Cc + k)
    pad = (k - 1) // 2
    st = torch.float16 if half else torch.float32
    tol = 2e-3 if half else 2e-5

    def rnd(*s, scale=1.0):
        v = torch.randn(*s, generator=g) * scale
        return v.half().float() if half else v
    x = rnd(N, H, W, Cc)
    w = torch.randn(Cc, 1, k, k, generator=g) * 0.3
    gamma, beta = 1.0 + 0.2 * torch.randn(Cc, generator=g), 0.1 * torch.randn(Cc, generator=g)
    dy = rnd(N, H, W, Cc)
    add = rnd(N, H, W, Cc)
    alpha = torch.tensor(0.4)
    # ---- reference (float64)
    xd = x.double()
    mean, var = xd.mean((0, 1, 2)), xd.var((0, 1, 2), unbiased=False)
    xh = (xd - mean) / torch.sqrt(var + 1e-3)
    z = xh * gamma.double() + beta.double()
    a = (z * torch.sigmoid(z)).requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y_ref = F.conv2d(a.permute(0, 3, 1, 2), wd, padding=pad, groups=Cc).permute(0, 2, 3, 1)
    (y_ref * dy.double()).sum().backward()
    gate = float(1 - torch.sigmoid(alpha))
    da_ref = gate * a.grad + add.double()
    sg = torch.sigmoid(z)
    dz_ref = da_ref * (sg * (1 + z * (1 - sg)))
    # ---- device
    xg, dyg, addg = x.to(dev, st), dy.to(dev, st), add.to(dev, st)
    wt = w.view(Cc, k * k).t().contiguous().to(dev)
    acc = K.zeros64(2 * Cc, xg)
    K.colstats(xg.view(-1, Cc), acc)
    bn = K.DeferredBN(acc, Cc, N * H * W, gamma.to(dev), beta.to(dev), 1e-3, 1)
    stats = K.zeros64(2 * Cc, xg)
    y = K.dwtile_fwd(xg, wt, k, pad, pad, H, W, bn=bn, stats=stats)
    assert y.dtype == st and _rel(y, y_ref) < tol
    yd = y.double().cpu()
    assert _rel(stats[:Cc], yd.sum((0, 1, 2))) < 1e-9 and _rel(stats[Cc:], (yd * yd).sum((0, 1, 2))) < 1e-9
    y_plain = K.dwtile_fwd(xg, wt, k, pad, pad, H, W)
    assert _rel(y_plain, F.conv2d(xd.permute(0, 3, 1, 2), wd.detach(), padding=pad, groups=Cc).permute(0, 2, 3, 1)) < tol
    sacc = K.zeros64(2 * Cc, xg)
    dz = K.dwtile_bwd_data(dyg, wt, k, pad, pad, H, W, alpha.to(dev), 2, addg, xg, bn, sacc)
    assert _rel(dz, dz_ref) < tol
    dzd = dz.double().cpu()
    stol = 3e-3 if half else 2e-5
    assert _rel(sacc[:Cc], dz_ref.sum((0, 1, 2))) < stol and _rel(sacc[Cc:], (dz_ref * xh).sum((0, 1, 2))) < stol
    assert _rel(sacc[:Cc], dzd.sum((0, 1, 2))) < 1e-6            # the sums are those of the stored values
    da = K.dwtile_bwd_data(dyg, wt, k, pad, pad, H, W, alpha.to(dev), 2, addg)
    assert _rel(da, da_ref) < tol
    dw = K.dwtile_bwd_weight(xg, dyg, k, pad, pad, bn=bn, gate_alpha=alpha.to(dev), gate_mode=2)
    assert _rel(dw.view(Cc, k, k), gate * wd.grad.view(Cc, k, k)) < (1e-3 if half else 2e-5)
    # ---- round 5: both gradients from ONE kernel (dw_tile_bwd_kernel: both halo tiles staged once)
    sacc2 = K.zeros64(2 * Cc, xg)
    dz2, dw2 = K.dwtile_bwd(dyg, xg, wt, k, pad, pad, bn=bn, gate_alpha=alpha.to(dev), gate_mode=2, add=addg, sacc=sacc2)
    assert dz2.dtype == st and _rel(dz2, dz_ref) < tol
    assert torch.equal(dz2, dz)                                   # the same arithmetic in the same order as ud_dwtile epi 2
    assert _rel(sacc2[:Cc], dz_ref.sum((0, 1, 2))) < stol and _rel(sacc2[Cc:], (dz_ref * xh).sum((0, 1, 2))) < stol
    assert _rel(sacc2[:Cc], dzd.sum((0, 1, 2))) < 1e-6
    assert _rel(dw2.view(Cc, k, k), gate * wd.grad.view(Cc, k, k)) < (1e-3 if half else 2e-5)
    # ... and without a deferred BatchNorm in front of the conv (x is the conv's input itself: MBConv blocks with expand ratio 1)
    wd2 = w.double().requires_grad_(True)
    (F.conv2d(xd.permute(0, 3, 1, 2), wd2, padding=pad, groups=Cc).permute(0, 2, 3, 1) * dy.double()).sum().backward()
    da2, dw3 = K.dwtile_bwd(dyg, xg, wt, k, pad, pad, gate_alpha=alpha.to(dev), gate_mode=2, add=addg)
    assert _rel(da2, da_ref) < tol and torch.equal(da2, da)
    assert _rel(dw3.view(Cc, k, k), gate * wd2.grad.view(Cc, k, k)) < (1e-3 if half else 2e-5)
    da3, _ = K.dwtile_bwd(dyg, xg, wt, k, pad, pad)               # no gate, nothing added
    assert _rel(da3, a.grad) < tol


@pytest.mark.parametrize("N,H,Cc,k,pad", [(2, 32, 24, 3, (0, 1, 0, 1)), (2, 16, 40, 5, (2, 2, 2, 2)), (1, 64, 8, 3, (0, 1, 0, 1)),
                                          (3, 16, 36, 5, (1, 2, 1, 2)), (2, 18, 12, 5, (1, 2, 1, 2)), (1, 21, 8, 3, (1, 1, 1, 1))])
@pytest.mark.parametrize("half", [False, True], ids=["fp32", "half"])
def test_tiled_depthwise_kernels_stride2(N, H, Cc, k, pad, half):
    """csrc/dwtile.hip at stride 2 with the static asymmetric pads of the four down-sampling MBConv blocks (pad = left, right,
    top, bottom: model/efficientnet/utils.py:264-275): forward (window stepped by 2) with the deferred BatchNorm and the output
    statistics, data gradient through the zero-stuffed tile with act'(bn(x)) and the BatchNorm sums, weight gradient — against
    autograd in float64."""
    import torch.nn.functional as F
    from unidefense_amd import kernels as K
    dev = _dev()
    K.reset_zero_pool()
    g = torch.Generator().manual_seed(N * 100 + H + Cc + k)
    pl, pr, pt, pb = pad
    st = torch.float16 if half else torch.float32
    tol = 2e-3 if half else 2e-5
    Ho, Wo = (H + pt + pb - k) // 2 + 1, (H + pl + pr - k) // 2 + 1

    def rnd(*s):
        v = torch.randn(*s, generator=g)
        return v.half().float() if half else v
    x, dy, add = rnd(N, H, H, Cc), rnd(N, Ho, Wo, Cc), rnd(N, H, H, Cc)
    w = torch.randn(Cc, 1, k, k, generator=g) * 0.3
    gamma, beta = 1.0 + 0.2 * torch.randn(Cc, generator=g), 0.1 * torch.randn(Cc, generator=g)
    xd = x.double()
    mean, var = xd.mean((0, 1, 2)), xd.var((0, 1, 2), unbiased=False)
    xh = (xd - mean) / torch.sqrt(var + 1e-3)
    z = xh * gamma.double() + beta.double()
    a = (z * torch.sigmoid(z)).requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y_ref = F.conv2d(F.pad(a.permute(0, 3, 1, 2), (pl, pr, pt, pb)), wd, stride=2, groups=Cc).permute(0, 2, 3, 1)
    assert y_ref.shape == (N, Ho, Wo, Cc)
    (y_ref * dy.double()).sum().backward()
    da_ref = a.grad + add.double()
    sg = torch.sigmoid(z)
    dz_ref = da_ref * (sg * (1 + z * (1 - sg)))
    xg, dyg, addg = x.to(dev, st), dy.to(dev, st), add.to(dev, st)
    wt = w.view(Cc, k * k).t().contiguous().to(dev)
    acc = K.zeros64(2 * Cc, xg)
    K.colstats(xg.view(-1, Cc), acc)
    bn = K.DeferredBN(acc, Cc, N * H * H, gamma.to(dev), beta.to(dev), 1e-3, 1)
    stats = K.zeros64(2 * Cc, xg)
    y = K.dwtile_fwd(xg, wt, k, pt, pl, Ho, Wo, bn=bn, stats=stats, stride=2)
    assert y.dtype == st and _rel(y, y_ref) < tol
    yd = y.double().cpu()
    assert _rel(stats[:Cc], yd.sum((0, 1, 2))) < 1e-9 and _rel(stats[Cc:], (yd * yd).sum((0, 1, 2))) < 1e-9
    sacc = K.zeros64(2 * Cc, xg)
    dz = K.dwtile_bwd_data(dyg, wt, k, pt, pl, H, H, None, 0, addg, xg, bn, sacc, stride=2)
    assert _rel(dz, dz_ref) < tol
    stol = 3e-3 if half else 2e-5
    assert _rel(sacc[:Cc], dz_ref.sum((0, 1, 2))) < stol and _rel(sacc[Cc:], (dz_ref * xh).sum((0, 1, 2))) < stol
    da = K.dwtile_bwd_data(dyg, wt, k, pt, pl, H, H, None, 0, addg, stride=2)
    assert _rel(da, da_ref) < tol
    dw = K.dwtile_bwd_weight(xg, dyg, k, pt, pl, bn=bn, stride=2)
    assert _rel(dw.view(Cc, k, k), wd.grad.view(Cc, k, k)) < (1e-3 if half else 2e-5)


@pytest.mark.parametrize("two_pass", [False, True], ids=["one-kernel", "two-pass"])
def test_producers_report_the_absmax_of_what_they_wrote(two_pass):
    """The planes GEMM (ud_gemm_p3 prec 2) scales an operand by the power of two its largest magnitude asks for.  The kernels that
    WRITE the operands (ud_se_scale_bn, ud_residual_bn, ud_normbwd_apply, ud_rfft2_ex / ud_rfft2_two_pass) leave that maximum in
    256 slots as a side output, so ud_absmax need not read the tensor again: the slots' maximum must be max|out| to the bit, on
    shapes whose last workgroups are partial, and split_planes given the slots must produce the planes it produces without."""
    from unidefense_amd import kernels as K
    from unidefense_amd.config import override
    dev = _dev()
    K.reset_zero_pool()

    def check(out, what):
        slots = out._ud_absmax
        assert slots is not None and slots.numel() == 256, what
        torch.cuda.synchronize()
        assert slots.max().item() == out.abs().max().item(), what
        o2 = out.reshape(-1, out.shape[-1])
        if o2.shape[0] % 32 == 0:
            a, b = K.split_planes(o2, prec=2, absmax=slots), K.split_planes(o2, prec=2)
            torch.cuda.synchronize()
            rows = lambda pl: pl.buf.view(2, pl.npanel, pl.panel // 32, 32)[:, :, :o2.shape[0]]     # (backing rows beyond R: unwritten)
            assert torch.equal(rows(a), rows(b)) and torch.equal(a.inv, b.inv), what

    with override(spectral_p2="on"):
        for G, R, Cc in ((3, 1000, 40), (4, 256, 672), (2, 4096, 24)):
            g, x, gamma, beta = _mk(G, R, Cc, 5 + Cc, dev)
            x = x * 37.0
            bn = _deferred(K, x, gamma, beta, 1e-3, 1)
            s = torch.randn(G, Cc, generator=g).to(dev)
            check(K.se_scale_bn(x, bn, s, G, R, want_absmax=True), "se_scale_bn")
            assert getattr(K.se_scale_bn(x, bn, s, G, R), "_ud_absmax", None) is None
            bn0 = _deferred(K, x, gamma, beta, 1e-3, 0)
            keep = (torch.rand(G, generator=g) < 0.7).float().to(dev)
            skip = torch.randn(G, R, Cc, generator=g).to(dev)
            check(K.residual_bn(x, bn0, keep, 1.25, skip, G, R, want_absmax=True), "residual_bn + skip")
            check(K.residual_bn(x, bn0, keep, 1.25, None, G, R, want_absmax=True), "residual_bn")
            dy = torch.randn(G, R, Cc, generator=g).to(dev) * 1e-3
            for act, b in ((1, bn), (0, bn0)):
                sacc = K.zeros64(2 * Cc, x)
                K.normbwd_sums(x, dy, keep, 1.25, b, False, G, R, sacc)
                check(K.normbwd_apply(x, dy, keep, 1.25, b, False, G, R, sacc, want_absmax=True)[0], "normbwd_apply")
        saved = K._FFT_TWO_PASS
        K._FFT_TWO_PASS = two_pass
        try:
            for S, Cc, N in ((64, 144, 3), (32, 200, 2), (16, 40, 4), (8, 272, 5)):
                g = torch.Generator().manual_seed(S + Cc)
                x = torch.randn(N, S, S, Cc, generator=g).to(dev) * 11.0
                gamma, beta = (1.0 + 0.2 * torch.randn(Cc, generator=g)).to(dev), (0.1 * torch.randn(Cc, generator=g)).to(dev)
                alpha = torch.tensor(-0.3, device=dev)
                bn = _deferred(K, x.view(N, S * S, Cc), gamma, beta, 1e-3, 1)
                check(K.rfft2(x, 1.0 / S, 2.0, want_absmax=True), "rfft2")
                check(K.rfft2_ex(x, 1.0 / S, 1.0, bn=bn, want_act=True, want_absmax=True)[0], "rfft2_ex bn")
                slots = torch.randn(64, generator=g, dtype=torch.float64).to(dev)
                check(K.rfft2_ex(x, 1.0 / S, 2.0, gate_alpha=alpha, gate_mode=1, gate_acc=slots, want_absmax=True)[0], "rfft2_ex gate")
        finally:
            K._FFT_TWO_PASS = saved
    # half storage and the planes path switched off: no slots, the kernels run as before
    with override(spectral_p2="off"):
        g, x, gamma, beta = _mk(2, 256, 48, 7, dev)
        bn = _deferred(K, x, gamma, beta, 1e-3, 1)
        assert K.se_scale_bn(x, bn, torch.zeros(2, 48, device=dev), 2, 256, want_absmax=True)._ud_absmax is None


@pytest.mark.parametrize("half", [False, True], ids=["fp32", "half"])
@pytest.mark.parametrize("Cc,N", [(336, 3), (192, 2), (40, 2), (36, 1)])
def test_fft32_lane_pair_form_equals_one_lane_form(Cc, N, half):
    """S = 32: the transform shared by a lane pair (even / odd decimated samples on lanes l and l ^ 32, the last radix-2 stage
    across the halves by wavefront shuffles: rfft2_wave_kernel / irfft2_wave_kernel) runs the same butterflies on the same values
    as the one-lane-per-row kernels: every output agrees to the last bit or two, with every fused prologue / epilogue (deferred
    BN + swish + activated copy + running statistics, gate factor and gate gradient, |Y|max slots; SF mix + freq - spat + BN
    sums), channel counts that do not fill the 32-channel groups, both storage types.  Against float64:
    test_fft_fused_variants / test_half_storage_kernels (whichever form the policy picks)."""
    from unidefense_amd import kernels as K
    from unidefense_amd import lib
    from unidefense_amd.config import override
    dev = _dev()
    K.reset_zero_pool()
    S = 32
    dt = torch.float16 if half else torch.float32
    g = torch.Generator().manual_seed(S + Cc)
    x = torch.randn(N, S, S, Cc, generator=g).to(dev).to(dt)
    gamma, beta = (1.0 + 0.2 * torch.randn(Cc, generator=g)).to(dev), (0.1 * torch.randn(Cc, generator=g)).to(dev)
    alpha = torch.tensor(-0.3, device=dev)
    Yin = torch.randn(N, S, S // 2 + 1, 2 * Cc, generator=g).to(dev).to(dt)
    spat = torch.randn(N, S, S, Cc, generator=g).to(dev).to(dt)
    slots = torch.randn(64, generator=g, dtype=torch.float64).to(dev)

    def run(mode):
        prev = lib.call("ud_fft32_set_wave", mode)
        saved = K._FFT_TWO_PASS
        K._FFT_TWO_PASS = False
        try:
            with override(spectral_p2="on"):
                out = {}
                out["rfft"] = K.rfft2(x, 1.0 / S, 2.0)
                out["irfft"] = K.irfft2(Yin, 1.0 / S, 0.5)
                rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
                bn = _deferred(K, x.view(N, S * S, Cc), gamma, beta, 1e-3, 1, rm, rv, 0.01)
                out["ex Y"], out["ex act"] = K.rfft2_ex(x, 1.0 / S, 1.0, bn=bn, want_act=True, update=True, want_absmax=not half)
                if not half:
                    out["ex |Y|max"] = out["ex Y"]._ud_absmax.max().reshape(1)
                out["ex running_mean"], out["ex running_var"] = rm, rv
                out["gate Y"], _, out["gate grad"] = K.rfft2_ex(x, 1.0 / S, 2.0, gate_alpha=alpha, gate_mode=1, gate_acc=slots)
                acc = K.zeros64(2 * Cc, x)
                out["mix y"], out["mix diff"] = K.irfft2_mix(Yin, 1.0 / S, spat, alpha, acc)
                torch.cuda.synchronize()
                return out, acc.clone()
        finally:
            K._FFT_TWO_PASS = saved
            lib.call("ud_fft32_set_wave", prev)
    o1, a1 = run(1)
    o2, a2 = run(2)
    ulp = 2.0 ** -10 if half else 2.0 ** -23
    for k in o1:
        scale = float(o1[k].double().abs().max())
        e = float((o1[k].double() - o2[k].double()).abs().max()) / (scale if scale > 0 else 1.0)
        assert e <= 2.0 * ulp, (k, e)
    assert _rel(a2, a1) < (1e-3 if half else 1e-6)


@pytest.mark.parametrize("N,HW,Cc", [(2, 64, 96), (3, 100, 64), (32, 64, 1632)])
def test_se_scale_writes_the_project_planes_itself(N, HW, Cc):
    """ud_colsum_bn_amax + ud_se_scale_bn_planes (round 5): the SE squeeze pass leaves max |swish(bn1(d))| behind, and the gate pass
    writes c = swish(bn1(d)) * sigmoid(s) straight into the project conv's fp16 x 2 planes with the scale that maximum gives.  The
    pooled sums equal ud_colsum_bn's; the planes re-assemble to ud_se_scale_bn's fp32 result (2^-21 of the maximum, the scale at
    most two binades above the exact one); the slots hold exactly max |swish(bn1(d))|."""
    from unidefense_amd import kernels as K
    dev = _dev()
    K.reset_zero_pool()
    g = torch.Generator().manual_seed(N + HW + Cc)
    x = torch.randn(N, HW, Cc, generator=g).to(dev)
    s = torch.randn(N, Cc, generator=g).to(dev)
    gamma, beta = (1.0 + 0.3 * torch.randn(Cc, generator=g)).to(dev), (0.2 * torch.randn(Cc, generator=g)).to(dev)
    acc = K.zeros64(2 * Cc, x)
    K.colstats(x.view(-1, Cc), acc)
    bn = K.DeferredBN(acc, Cc, N * HW, gamma, beta, 1e-3, 1)
    pool_ref = K.zeros64(N * Cc, x)
    K.colsum_bn(x, bn, N, HW, pool_ref)
    pool = K.zeros64(N * Cc, x)
    amax = K.colsum_bn_amax(x, bn, N, HW, pool)
    y = K.se_scale_bn(x, bn, s, N, HW)
    pl = K.se_scale_bn_planes(x, bn, s, N, HW, amax)
    a = K.bn_apply(x, bn, N, HW)
    torch.cuda.synchronize()
    assert torch.equal(pool, pool_ref)
    assert float(amax.view(torch.float32).max()) == float(a.abs().max())
    R = N * HW
    h = pl.buf.view(2, pl.npanel, pl.panel // 32, 32)[:, :, :R].view(torch.float16).permute(0, 2, 1, 3).reshape(2, R, Cc).double()
    inv = float(pl.inv)
    yd = y.view(R, Cc).double()
    top = float(yd.abs().max())
    loose = 2.0 ** 15 / (top / inv)
    assert 1.0 <= loose < 8.0, loose
    assert float(((h[0] + h[1] / 2048.0) * inv - yd).abs().max()) <= 2.0 ** -21 * top * loose


@pytest.mark.parametrize("N,S,Cc,k", [(2, 8, 64, 5), (3, 8, 40, 3), (32, 8, 96, 5), (2, 16, 32, 5), (3, 16, 48, 3), (8, 16, 96, 5)])
def test_adjoint_transform_does_the_depthwise_backward(N, S, Cc, k):
    """ud_irfft2_dwbwd (csrc/fft.hip, round 5; the 8 x 8 maps): ONE kernel = ud_irfft2 (adjoint of rfft2) + the depthwise data
    gradient with gate, the added spectral-branch gradient, act'(bn(x)) and the BatchNorm backward sums + the depthwise weight
    gradient.  Against the separate kernels (themselves held to float64 autograd above): dz and the sums to 1e-6, the weight
    gradient to 2e-6 of its scale."""
    from unidefense_amd import kernels as K
    dev = _dev()
    K.reset_zero_pool()
    g = torch.Generator().manual_seed(N * 10 + Cc + k)
    x = torch.randn(N, S, S, Cc, generator=g).to(dev)
    dd = torch.randn(N, S, S, Cc, generator=g).to(dev)
    Yf = torch.randn(N, S, S // 2 + 1, 2 * Cc, generator=g).to(dev)
    wt = (0.3 * torch.randn(k * k, Cc, generator=g)).to(dev)
    gamma, beta = (1.0 + 0.3 * torch.randn(Cc, generator=g)).to(dev), (0.2 * torch.randn(Cc, generator=g)).to(dev)
    alpha = torch.tensor([0.4], device=dev)
    acc = K.zeros64(2 * Cc, x)
    K.colstats(x.view(-1, Cc), acc)
    bn = K.DeferredBN(acc, Cc, N * S * S, gamma, beta, 1e-3, 1)
    pad = (k - 1) // 2
    da_f = K.irfft2(Yf, 1.0 / S, 0.5)
    s_ref = K.zeros64(2 * Cc, x)
    dz_ref, dw_ref = K.dwtile_bwd(dd, x, wt, k, pad, pad, bn=bn, gate_alpha=alpha, gate_mode=2, add=da_f, sacc=s_ref)
    s_new = K.zeros64(2 * Cc, x)
    dz, dw = K.irfft2_dwbwd(Yf, 1.0 / S, 0.5, dd, x, bn, wt, k, alpha, 2, s_new)
    # a 3C accumulator also receives the energy of dz per channel, rounded up (ud_normbwd_apply_planes' bound)
    s_en = K.zeros64(3 * Cc, x)
    K.irfft2_dwbwd(Yf, 1.0 / S, 0.5, dd, x, bn, wt, k, alpha, 2, s_en)
    torch.cuda.synchronize()
    assert _rel(dz, dz_ref.double().cpu()) < 2e-6
    assert _rel(s_new, s_ref.cpu()) < 1e-6
    assert _rel(dw, dw_ref.double().cpu()) < 2e-6
    assert _rel(s_en[:2 * Cc], s_ref.cpu()) < 1e-6
    en_ref = dz.double().pow(2).sum((0, 1, 2))
    ratio = (s_en[2 * Cc:] / en_ref).cpu()
    assert float(ratio.min()) >= 1.0 and float(ratio.max()) < 1.001, (float(ratio.min()), float(ratio.max()))


def test_depthwise_weight_gradient_folds_deferred_to_one_launch():
    """ud_dwtile_wgrad_finalize_multi (round 5): between kernels.begin_wgrad_folds() and flush_wgrad_folds() the depthwise weight
    gradients of ud_dwtile_bwd / ud_dwtile_wgrad / ud_irfft2_dwbwd stay partial rows in buffers of their own (the entry points return
    the row count), and ONE launch folds them all — bit for bit the per-conv folds (same kernel body, same order of the rows)."""
    from unidefense_amd import kernels as K
    from unidefense_amd.config import override
    dev = _dev()
    K.reset_zero_pool()
    g = torch.Generator().manual_seed(5)
    alpha = torch.tensor([0.3], device=dev)

    def run():
        out = []
        for (N, S, Cc, k) in ((2, 16, 64, 3), (3, 32, 40, 5), (4, 8, 96, 5)):
            x = torch.randn(N, S, S, Cc, generator=g).to(dev)
            dy = torch.randn(N, S, S, Cc, generator=g).to(dev)
            wt = (0.3 * torch.randn(k * k, Cc, generator=g)).to(dev)
            gamma, beta = (1.0 + 0.3 * torch.randn(Cc, generator=g)).to(dev), (0.2 * torch.randn(Cc, generator=g)).to(dev)
            acc = K.zeros64(2 * Cc, x)
            K.colstats(x.view(-1, Cc), acc)
            bn = K.DeferredBN(acc, Cc, N * S * S, gamma, beta, 1e-3, 1)
            pad = (k - 1) // 2
            out.append(K.dwtile_bwd(dy, x, wt, k, pad, pad, bn=bn, gate_alpha=alpha, gate_mode=2, sacc=K.zeros64(2 * Cc, x))[1])
            out.append(K.dwtile_bwd_weight(x, dy, k, pad, pad, bn=bn, gate_alpha=alpha, gate_mode=1))
            out.append(K.dwtile_bwd(dy, x, wt, k, pad, pad)[1])
            if S in (8, 16):
                Yf = torch.randn(N, S, S // 2 + 1, 2 * Cc, generator=g).to(dev)
                out.append(K.irfft2_dwbwd(Yf, 1.0 / S, 0.5, dy, x, bn, wt, k, alpha, 2, K.zeros64(2 * Cc, x))[1])
        return out

    with override(deterministic=True):          # ordered sums everywhere: the two runs differ in the folds' launches only
        g.manual_seed(5)
        ref = run()
        g.manual_seed(5)
        K.begin_wgrad_folds()
        got = run()
        assert all(getattr(t, "_ud_deferred", False) for t in got) and len(K._WGRAD_FOLDS) == len(got)
        K.flush_wgrad_folds(end=True)
    torch.cuda.synchronize()
    assert K._WGRAD_FOLDS is None and len(ref) == 11
    for a, b in zip(got, ref):
        assert torch.equal(a, b)


@pytest.mark.parametrize("N,HW,Cc,act,dz", [(2, 64, 96, 0, False), (3, 100, 40, 1, True), (32, 64, 272, 0, False), (4, 256, 960, 1, True)])
def test_normbwd_apply_writes_the_gemm_planes_itself(N, HW, Cc, act, dz):
    """ud_normbwd_sums' third sum + ud_normbwd_apply_planes (round 5): the BatchNorm backward in front of a 1x1 conv writes its
    result straight into the fp16 x 2 planes the conv's weight / data gradient GEMMs read, scaled by the a-priori bound
    max_c |gamma_c invstd_c| sqrt(sum dz_c^2).  The sums equal the two-sum launch's; the energy is sum dz^2 rounded up; the planes
    re-assemble to ud_normbwd_apply's fp32 result to 2^-21 of the scale; the bound holds and is within 2^7 of the exact maximum;
    the pad columns of the last 32-wide panel are zero (they multiply as k in the data gradient)."""
    from unidefense_amd import kernels as K
    dev = _dev()
    K.reset_zero_pool()
    g = torch.Generator().manual_seed(N + HW + Cc)
    x = torch.randn(N, HW, Cc, generator=g).to(dev)
    dy = torch.randn(N, HW, Cc, generator=g).to(dev) * torch.rand(Cc, generator=g).to(dev)
    keep = (torch.rand(N, generator=g) < 0.8).float().to(dev) if not dz else None
    inv_keep = 1.25 if not dz else 1.0
    gamma, beta = (1.0 + 0.3 * torch.randn(Cc, generator=g)).to(dev), (0.2 * torch.randn(Cc, generator=g)).to(dev)
    acc = K.zeros64(2 * Cc, x)
    K.colstats(x.view(-1, Cc), acc)
    bn = K.DeferredBN(acc, Cc, N * HW, gamma, beta, 1e-3, act)
    s2 = K.zeros64(2 * Cc, x)
    K.normbwd_sums(x, dy, keep, inv_keep, bn, dz, N, HW, s2)
    s3 = K.zeros64(3 * Cc, x)
    K.normbwd_sums(x, dy, keep, inv_keep, bn, dz, N, HW, s3)
    ref, dg_ref, db_ref = K.normbwd_apply(x, dy, keep, inv_keep, bn, dz, N, HW, s2)
    pl, dg, db = K.normbwd_apply_planes(x, dy, keep, inv_keep, bn, dz, N, HW, s3)
    # the incoming gradient as the kernels see it (dz): from the fp32 kernel with gamma = invstd-free identity is not available, so
    # restate it: dy * keep * inv_keep * act'(bn(x))
    xd = x.double()
    mean, var = xd.mean((0, 1)), xd.var((0, 1), unbiased=False)
    xh = (xd - mean) / torch.sqrt(var + 1e-3)
    z = gamma.double() * xh + beta.double()
    gin = dy.double()
    if not dz:
        gin = gin * (keep.double() * inv_keep).view(N, 1, 1)
        if act:
            sg = torch.sigmoid(z)
            gin = gin * (sg * (1 + z * (1 - sg)))
    torch.cuda.synchronize()
    assert _rel(s3[:2 * Cc], s2.cpu()) < 1e-12
    ratio = (s3[2 * Cc:] / gin.pow(2).sum((0, 1))).cpu()
    assert float(ratio.min()) >= 1.0 - 1e-6 and float(ratio.max()) < 1.001, (float(ratio.min()), float(ratio.max()))
    assert _rel(dg, dg_ref.double().cpu()) < 1e-6 and _rel(db, db_ref.double().cpu()) < 1e-6
    R = N * HW
    full = pl.buf.view(2, pl.npanel, pl.panel // 32, 32)[:, :, :R].view(torch.float16).permute(0, 2, 1, 3).reshape(2, R, pl.npanel * 32)
    h = full[:, :, :Cc].double()
    assert float(full[:, :, Cc:].abs().max() if pl.npanel * 32 > Cc else 0.0) == 0.0
    inv = float(pl.inv)
    yd = ref.view(R, Cc).double()
    top = float(yd.abs().max())
    loose = 2.0 ** 15 / (top / inv)
    assert 1.0 <= loose < 128.0, loose
    assert float(((h[0] + h[1] / 2048.0) * inv - yd).abs().max()) <= 2.0 ** -21 * top * loose


@pytest.mark.parametrize("N,HW,Cc,skip,keep_p", [(32, 256, 160, True, 0.9), (32, 64, 272, True, 1.0), (4, 256, 112, False, 1.0),
                                                  (3, 50, 36, True, 0.8)])
def test_residual_bn_writes_the_next_expand_planes(N, HW, Cc, skip, keep_p):
    """ud_residual_bn_planes (round 6): BN2 + drop-connect + skip also written as the fp16 x 2 planes of the NEXT block's expand
    conv, scaled by the a-priori bound max_c(|gamma_c| sqrt(count) + |beta_c|) / keep_prob + max |skip|.  The fp32 result and its
    absmax slots are those of ud_residual_bn bit for bit; the planes re-assemble to it to 2^-21 of the scale; the bound holds and is
    within 2^8 of the exact maximum; pad columns of the last panel are zero."""
    from unidefense_amd import kernels as K
    from unidefense_amd.lib import call as _call
    import ctypes as C
    dev = _dev()
    K.reset_zero_pool()
    g = torch.Generator().manual_seed(N + HW + Cc)
    p = (torch.randn(N, HW, Cc, generator=g) * (0.5 + torch.rand(Cc, generator=g))).to(dev)
    sk = None
    if skip:
        prev = torch.randn(N, HW, Cc, generator=g).to(dev) * 2.0
        acc0 = K.zeros64(2 * Cc, p)
        K.colstats(prev.view(-1, Cc), acc0)
        bn0 = K.DeferredBN(acc0, Cc, N * HW, torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev), 1e-3, 0)
        sk = K.residual_bn(prev, bn0, None, 1.0, None, N, HW, want_absmax=True)          # a skip with its producer's absmax slots
    keep = (torch.rand(N, generator=g) < keep_p).float().to(dev) if keep_p < 1.0 else None
    gamma, beta = (1.0 + 0.3 * torch.randn(Cc, generator=g)).to(dev), (0.2 * torch.randn(Cc, generator=g)).to(dev)
    acc = K.zeros64(2 * Cc, p)
    K.colstats(p.view(-1, Cc), acc)
    bn = K.DeferredBN(acc, Cc, N * HW, gamma, beta, 1e-3, 0)
    ref = K.residual_bn(p, bn, keep, 1.0 / keep_p, sk, N, HW, want_absmax=True)
    out = torch.empty_like(p)
    amax = K.amax_slots(p, True)
    pl = K.Planes(N * HW, Cc, p, 2, False)
    _call("ud_residual_bn_planes", K._p(p), C.byref(bn.ref()), K._p(keep), float(1.0 / keep_p), K._p(sk),
          K._p(sk._ud_absmax if sk is not None else None), K._p(out), K._p(pl.buf), pl.panel, pl.plane, K._p(pl.inv), N, HW, Cc,
          K._p(amax), K._stream())
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert float(amax.view(torch.float32).max()) == float(ref._ud_absmax.view(torch.float32).max()) == float(ref.abs().max())
    R = N * HW
    full = pl.buf.view(2, pl.npanel, pl.panel // 32, 32)[:, :, :R].view(torch.float16).permute(0, 2, 1, 3).reshape(2, R, pl.npanel * 32)
    h = full[:, :, :Cc].double()
    assert float(full[:, :, Cc:].abs().max() if pl.npanel * 32 > Cc else 0.0) == 0.0
    inv = float(pl.inv)
    yd = ref.view(R, Cc).double()
    top = float(yd.abs().max())
    loose = 2.0 ** 15 / (top / inv)
    assert within("residual planes: bound / exact maximum", loose, 256.0) and loose >= 1.0, loose
    assert within("residual planes: re-assembled error / (2^-21 scale)", float(((h[0] + h[1] / 2048.0) * inv - yd).abs().max()) /
                  (2.0 ** -21 * top * loose), 1.0)


@pytest.mark.parametrize("M,Ce,Cin,with_add", [(32 * 40, 144, 24, False), (32 * 7 + 5, 144, 24, True), (3001, 192, 32, True),
                                                (32 * 600 + 17, 192, 32, False), (131072, 144, 24, True)])
def test_expand_conv_backward_in_one_pass(M, Ce, Cin, with_add):
    """ud_pw_bwd_fused (round 6): BatchNorm-0 backward applied on load + the thin expand conv's weight and data gradient from one
    LDS image, against float64 (model/efficientnet/model.py:101-109 differentiated) and against the three launches it replaces
    (ud_normbwd_apply + gemm_tn + gemm_nn).  Ragged row counts (a last tile of 5 / 25 / 17 rows), fewer tiles than workgroups and
    more (the strided tile loop), with and without the skip path's gradient added in place."""
    from unidefense_amd import kernels as K
    dev = _dev()
    K.reset_zero_pool()
    g = torch.Generator().manual_seed(M % 1000 + Ce)
    x = torch.randn(M, Cin, generator=g).to(dev)
    w = (torch.randn(Ce, Cin, generator=g) / Cin ** 0.5).to(dev)
    e = (x @ w.t()).contiguous()
    dz = (torch.randn(M, Ce, generator=g) * (0.2 + torch.rand(Ce, generator=g))).to(dev)
    gamma, beta = (1.0 + 0.3 * torch.randn(Ce, generator=g)).to(dev), (0.2 * torch.randn(Ce, generator=g)).to(dev)
    skip = torch.randn(M, Cin, generator=g).to(dev) if with_add else None
    acc = K.zeros64(2 * Ce, x)
    K.colstats(e, acc)
    bn = K.DeferredBN(acc, Ce, M, gamma, beta, 1e-3, 1)
    sb = K.zeros64(2 * Ce, x)
    K.normbwd_sums(e.view(1, M, Ce), dz.view(1, M, Ce), None, 1.0, bn, True, 1, M, sb)
    assert K.expand_bwd_fused_ok(e, w, True) and not K.expand_bwd_fused_ok(e, w, False)
    # the three launches
    de, dg_ref, db_ref = K.normbwd_apply(e.view(1, M, Ce), dz.view(1, M, Ce), None, 1.0, bn, True, 1, M, sb)
    dw_ref = K.gemm_tn(de.view(M, Ce), x)
    dx_ref = K.gemm_nn(de.view(M, Ce), w)
    if with_add:
        dx_ref = dx_ref + skip
    add = skip.clone() if with_add else None
    dx, dw, dg, db = K.expand_bwd_fused(e, dz, bn, sb, None, x, w, add=add)
    if with_add:
        assert dx.data_ptr() == add.data_ptr()
    # float64
    ed, dzd = e.double(), dz.double()
    mean, var = ed.mean(0), ed.var(0, unbiased=False)
    inv = 1.0 / torch.sqrt(var + 1e-3)
    xh = (ed - mean) * inv
    ded = gamma.double() * inv * (dzd - dzd.mean(0) - xh * (dzd * xh).mean(0))
    dw64 = ded.t() @ x.double()
    dx64 = ded @ w.double()
    if with_add:
        dx64 = dx64 + skip.double()
    torch.cuda.synchronize()
    r_dx, r_dw = _rel(dx, dx64), _rel(dw, dw64)
    r_dx0, r_dw0 = _rel(dx_ref, dx64), _rel(dw_ref, dw64)
    assert within("pw_bwd dx vs float64", r_dx, max(2e-6, 2 * r_dx0))
    assert within("pw_bwd dw vs float64", r_dw, max(2e-6, 2 * r_dw0))
    assert _rel(dg, dg_ref) < 1e-6 and _rel(db, db_ref) < 1e-6
    # bit-reproducible (partials folded in workgroup order)
    add2 = skip.clone() if with_add else None
    dx2, dw2, _, _ = K.expand_bwd_fused(e, dz, bn, sb, None, x, w, add=add2)
    assert torch.equal(dx2, dx) and torch.equal(dw2, dw)


@pytest.mark.parametrize("N,HW,Ce,Co,act", [(2, 64, 144, 32, 1), (3, 32 * 9, 192, 32, 1), (5, 4096, 192, 32, 1), (32, 4096, 144, 32, 1),
                                            (2, 96, 192, 32, 0), (2, 64, 336, 56, 1), (32, 1024, 336, 56, 1), (3, 1024, 192, 56, 1),
                                            (7, 32 * 5, 336, 56, 0), (3, 4096, 48, 24, 1), (2, 16384, 24, 24, 1), (5, 96, 24, 24, 0)])
def test_project_conv_backward_without_its_data_gradient(N, HW, Ce, Co, act, monkeypatch):
    """ud_pj_bwd_fused_a / _b (round 6): the thin project conv's backward with dc = dp Wp re-made per 32-row tile inside the two
    passes over d — weight gradient + SE dot; gate / swish backward + BatchNorm-1 sums; the 336- and 192-channel tensors in front of
    a 56-channel output walked as column chunks (112 / 96 wide, CO padded to two MFMA k-steps) — against float64
    (model/efficientnet/model.py:113-126 differentiated) and against the four launches they replace (gemm_tn, gemm_nn,
    ud_coldot_bn, ud_se_scale_bwd_bn).  One sample per workgroup chunk and several, fewer tiles than workgroups and more."""
    from unidefense_amd import kernels as K
    from unidefense_amd.config import cfg
    monkeypatch.setattr(cfg, "project_fused_narrow", True)          # (the 24-channel outputs are built and tested, shipped off: config.py)
    dev = _dev()
    K.reset_zero_pool()
    g = torch.Generator().manual_seed(N * 7 + HW + Ce)
    M = N * HW
    d = (torch.randn(N, HW, Ce, generator=g) * 1.3 + 0.2).to(dev)
    w = (torch.randn(Co, Ce, generator=g) / Ce ** 0.5).to(dev)
    dp = (torch.randn(M, Co, generator=g) * (0.3 + torch.rand(Co, generator=g))).to(dev)
    s = torch.randn(N, Ce, generator=g).to(dev)
    dpool = torch.randn(N, Ce, generator=g).to(dev)
    gamma, beta = (1.0 + 0.3 * torch.randn(Ce, generator=g)).to(dev), (0.2 * torch.randn(Ce, generator=g)).to(dev)
    acc = K.zeros64(2 * Ce, d)
    K.colstats(d.view(M, Ce), acc)
    bn = K.DeferredBN(acc, Ce, M, gamma, beta, 1e-3, act)
    assert K.project_bwd_fused_ok(d, w, HW) and not K.project_bwd_fused_ok(d, w, HW + 4)
    # the launches they replace
    c = K.se_scale_bn(d, bn, s, N, HW)
    dw_ref = K.gemm_tn(dp, c.view(M, Ce))
    dc = K.gemm_nn(dp, w).view(N, HW, Ce)
    dg_ref = K.zeros64(N * Ce, d)
    K.coldot_bn(dc, d, bn, N, HW, dg_ref)
    sb_ref = K.zeros64(2 * Ce, d)
    dz_ref = K.se_scale_bwd_bn(dc, d, bn, s, dpool, 1.0 / HW, N, HW, sb_ref)
    # one pass each
    dgate = K.zeros64(N * Ce, d)
    dw = K.project_bwd_fused_a(d, bn, s, dp, w, N, HW, dgate)
    sb = K.zeros64(2 * Ce, d)
    dz = K.project_bwd_fused_b(d, bn, s, dpool, 1.0 / HW, dp, w, N, HW, sb)
    # float64
    dd = d.double()
    mean, var = dd.mean((0, 1)), dd.var((0, 1), unbiased=False)
    xh = (dd - mean) / torch.sqrt(var + 1e-3)
    z = gamma.double() * xh + beta.double()
    sg = torch.sigmoid(z)
    a = z * sg if act else z
    da = sg * (1 + z * (1 - sg)) if act else torch.ones_like(z)
    gate = torch.sigmoid(s.double()).view(N, 1, Ce)
    c64 = (a * gate).view(M, Ce)
    dc64 = (dp.double() @ w.double()).view(N, HW, Ce)
    dw64 = dp.double().t() @ c64
    dg64 = (dc64 * a).sum(1).view(-1)
    dz64 = (dc64 * gate + dpool.double().view(N, 1, Ce) / HW) * da
    s164, s264 = dz64.sum((0, 1)), (dz64 * xh).sum((0, 1))
    torch.cuda.synchronize()
    for name, got, ref, want in (("dw", dw, dw_ref, dw64), ("dgate", dgate, dg_ref, dg64), ("dz", dz, dz_ref, dz64),
                                 ("s1", sb[:Ce], sb_ref[:Ce], s164), ("s2", sb[Ce:], sb_ref[Ce:], s264)):
        r, r0 = _rel(got, want), _rel(ref, want)
        assert within("pj_bwd %s vs float64" % name, r, max(3e-6, 2 * r0)), (name, r, r0)


@pytest.mark.parametrize("N,HW,Ce,Co,act", [(2, 64, 144, 32, 1), (3, 32 * 9, 192, 32, 1), (32, 4096, 192, 32, 1), (5, 4096, 144, 32, 0),
                                            (3, 4096, 48, 24, 1), (2, 16384, 24, 24, 1), (5, 96, 24, 24, 0)])
def test_project_conv_forward_with_the_gate_applied_on_load(N, HW, Ce, Co, act, monkeypatch):
    """ud_pj_fwd_fused (round 6): p = (act(bn1(d)) sigmoid(s)) Wp^T in one pass over d + p's BatchNorm-2 statistics, against
    float64 (model/efficientnet/model.py:113-126) and against ud_se_scale_bn + gemm_nt + ud_colstats."""
    from unidefense_amd import kernels as K
    from unidefense_amd.config import cfg
    monkeypatch.setattr(cfg, "project_fused_narrow", True)
    dev = _dev()
    K.reset_zero_pool()
    g = torch.Generator().manual_seed(N * 11 + HW + Ce)
    M = N * HW
    d = (torch.randn(N, HW, Ce, generator=g) * 1.3 + 0.2).to(dev)
    w = (torch.randn(Co, Ce, generator=g) / Ce ** 0.5).to(dev)
    s = torch.randn(N, Ce, generator=g).to(dev)
    gamma, beta = (1.0 + 0.3 * torch.randn(Ce, generator=g)).to(dev), (0.2 * torch.randn(Ce, generator=g)).to(dev)
    acc = K.zeros64(2 * Ce, d)
    K.colstats(d.view(M, Ce), acc)
    bn = K.DeferredBN(acc, Ce, M, gamma, beta, 1e-3, act)
    c = K.se_scale_bn(d, bn, s, N, HW)
    p_ref = K.gemm_nt(c.view(M, Ce), w)
    st_ref = K.zeros64(2 * Co, d)
    K.colstats(p_ref, st_ref)
    st = K.zeros64(2 * Co, d)
    p, ctx = K.project_fwd_fused(d, bn, s, w, N, HW, stats=st)
    assert ctx.plans is None and ctx.x is None
    dd = d.double()
    mean, var = dd.mean((0, 1)), dd.var((0, 1), unbiased=False)
    z = gamma.double() * (dd - mean) / torch.sqrt(var + 1e-3) + beta.double()
    a = z * torch.sigmoid(z) if act else z
    p64 = (a * torch.sigmoid(s.double()).view(N, 1, Ce)).view(M, Ce) @ w.double().t()
    torch.cuda.synchronize()
    assert within("pj_fwd p vs float64", _rel(p, p64), max(3e-6, 2 * _rel(p_ref, p64)))
    assert within("pj_fwd sum vs float64", _rel(st[:Co], p64.sum(0)), max(3e-6, 2 * _rel(st_ref[:Co], p64.sum(0))))
    assert within("pj_fwd sumsq vs float64", _rel(st[Co:], p64.pow(2).sum(0)), max(3e-6, 2 * _rel(st_ref[Co:], p64.pow(2).sum(0))))
