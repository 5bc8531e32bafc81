"""GPU, SELF-comparison (runs last: nothing here is an oracle test): the fused MBConv path (tape.mbconv_fused: deferred
BatchNorm, csrc/fused.hip) against the operator-by-operator path it replaces (tape.conv1x1 / batchnorm_act / sfconv_dw /
squeeze_excite / residual — each tested against torch in float64 by tests/test_a_kernels_gpu.py and against the reference's
goldens by tests/test_c_model_gpu.py).  Reference: model/efficientnet/model.py:94-135, exp.py:46-65.

Both paths run the same parameters, inputs and dropout / drop-connect masks.  Bars:
  * outputs / features: 2e-4 of the tensor's largest magnitude; activation gradients 1e-3;
  * parameter-gradient TENSORS: 3e-4 of the tensor's largest entry + the 2e-5 floor of test_c_model_gpu (BN2's bias gradient
    is a sum that cancels to rounding noise: the next block's BatchNorm removes any per-channel shift of its input);
  * SCALAR gate gradients (24 sf_coef, fuse_coef): each is ONE global sum  sigmoid'(a) * sum dy * (freq - spat)  over up to
    3e6 terms of both signs, so its error is measured against the sum of the terms' MAGNITUDES (captured by the operator
    path in debug mode: tape.gate_cond), not against what is left of the sum after cancellation: 1e-5 of it (observed: 2e-7).
"""
import pytest
import torch

from oracle import param_fill
from tests import oracle_util as ou
from tests.margins import within

pytestmark = pytest.mark.gpu
GATE_COND_RTOL = 1e-5


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def _run(dev, fused, sf, n, seed, running, debug=True):
    from unidefense_amd.config import override
    from unidefense_amd.loss import LOSSES
    from unidefense_amd.model import load_model
    m = load_model("UDEB4")(extractor="efficientnet-b4", num_classes=2, drop_rate=0.5)
    param_fill.fill_module_(m, sf_coef=sf, fuse_coef=0.3)
    m = m.to(dev).train()
    x = param_fill.make_input(n, 256, seed).to(dev)
    tgt = param_fill.make_labels(n).to(dev)
    rng = ou.make_rng(n, 1, 0.5)
    rng = {k: ({i: v.to(dev) for i, v in val.items()} if isinstance(val, dict) else val.to(dev)) for k, val in rng.items()}
    with override(fused_mbconv=fused):
        try:
            LOSSES["aw_triplet"].n_real = n // 2
            m._debug_watch = debug
            out = m(x, rng=rng)
            ld = out["loss_dict"]
            loss = LOSSES["cross_entropy"](out["cls_out"], tgt) + 0.1 * (ld["freq_mask"].mean() + ld["spat_mask"].mean()) \
                + 0.1 * sum(LOSSES["aw_triplet"](f, tgt) for f in ld["triplet"]) + 0.3 * ld["factorization"].square().mean() \
                + out["rec"].square().mean()
            loss.backward()
        finally:
            LOSSES["aw_triplet"].n_real = None
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    feats = {k: v.detach().clone() for k, v in m._debug_feats.items()} if debug else {}
    fgrads = {k: v.detach().clone() for k, v in m._debug_tape.captured.items() if v is not None} if debug else {}
    bufs = {k: v.detach().clone() for k, v in m.named_buffers()} if running else {}
    cond = {}
    if debug:
        ids = {id(p): k for k, p in m.named_parameters()}
        cond = {ids[i]: v for i, v in m._debug_tape.gate_cond.items()}
    return loss.detach(), out, feats, grads, fgrads, bufs, cond


def _grad_bars(g0, g1, cond, label):
    """Per-tensor deviation of g1 from g0 against the bars of the header; returns the offenders."""
    bad = []
    worst_t, worst_s = 0.0, 0.0
    for k in g0:
        d = float((g1[k].double() - g0[k].double()).abs().max())
        ref = float(g0[k].double().abs().max())
        if g0[k].dim() == 0 and k in cond:
            tol = GATE_COND_RTOL * cond[k]
            worst_s = max(worst_s, d / tol)
        else:
            tol = 3e-4 * ref + 2e-5
            worst_t = max(worst_t, d / tol)
        if not d <= tol:
            bad.append(("grad " + k, d, ref, tol))
    ok_t = within(label + ": worst tensor-gradient deviation / bar", worst_t, 1.0)
    ok_s = within(label + ": worst gate-gradient deviation / (1e-5 x sum|terms|)", worst_s, 1.0)
    print(f"  {label}: worst deviation / bar: tensors {worst_t:.3f}, scalar gates {worst_s:.3f} ({len(g0)} gradients)")
    assert (ok_t and ok_s) == (not bad)
    return bad


@pytest.mark.parametrize("sf", [0.0, -10.0])
def test_fused_mbconv_equals_operator_path(sf):
    dev = _dev()
    l0, o0, f0, g0, fg0, b0, cond = _run(dev, False, sf, 4, 11, True)
    l1, o1, f1, g1, fg1, b1, _ = _run(dev, True, sf, 4, 11, True)
    assert within("loss", _rel(l1, l0), 1e-5)
    scalars = [k for k in g0 if g0[k].dim() == 0]
    assert len(scalars) == 25 and set(scalars) <= set(cond), sorted(set(scalars) - set(cond))
    bad = []
    for k in f0:
        if not within("feat " + k, _rel(f1[k], f0[k]), 2e-4):
            bad.append(("feat " + k, _rel(f1[k], f0[k])))
    for k in fg0:
        if not within("dfeat " + k, _rel(fg1[k], fg0[k]), 1e-3):
            bad.append(("dfeat " + k, _rel(fg1[k], fg0[k])))
    assert set(g0) == set(g1)
    bad += _grad_bars(g0, g1, cond, "fused vs operator")
    # running statistics and batch counters move identically
    worst_b = 0.0
    for k in b0:
        if b0[k].dtype.is_floating_point:
            e = _rel(b1[k], b0[k])
            worst_b = max(worst_b, e)
            if e > 1e-5:
                bad.append(("buffer " + k, e))
        else:
            assert torch.equal(b0[k], b1[k]), k
    within("running statistics", worst_b, 1e-5)
    assert not bad, bad[:20]


def test_skip_gradient_accumulated_in_place_equals_separate_add():
    """Without a debug watch the expand conv's data gradient is accumulated INTO the skip branch's gradient buffer
    (GEMM epilogue) instead of a separate add: same gradients."""
    dev = _dev()
    _, _, _, g0, _, _, cond = _run(dev, False, 0.0, 4, 3, False, debug=True)      # operator path: conditioning sums
    _, _, _, g1, _, _, _ = _run(dev, True, 0.0, 4, 3, False, debug=True)
    _, _, _, g2, _, _, _ = _run(dev, True, 0.0, 4, 3, False, debug=False)
    bad = _grad_bars(g1, g2, cond, "in-place skip gradient vs separate add")
    assert not bad, bad[:10]


@pytest.mark.parametrize("half", [False, True], ids=["fp32", "half"])
def test_tiled_depthwise_policy_equals_strip_kernels(half):
    """tape._dw_tile_policy routes some depthwise convs of the fused node to the LDS-tiled kernels (csrc/dwtile.hip); with
    the policy off every one of them runs on the strip kernels.  Same step either way: fp32 storage to fp32 rounding; half
    storage to a few fp16 roundings (the tiled forward reads swish(bn0(e)) unrounded, the strip kernel reads the fp16 copy
    rfft2_ex / bn_apply stored)."""
    from unidefense_amd import tape as T
    from unidefense_amd.loss import LOSSES
    from unidefense_amd.model import load_model
    dev = _dev()
    n = 8

    def run(tiled):
        saved = T._DW_TILED
        T._DW_TILED = tiled
        try:
            m = load_model("UDEB4")(extractor="efficientnet-b4", num_classes=2, drop_rate=0.5)
            param_fill.fill_module_(m, sf_coef=0.0, fuse_coef=0.3)
            m = m.to(dev).train()
            m.half_storage = half
            x = param_fill.make_input(n, 256, 31).to(dev)
            tgt = param_fill.make_labels(n).to(dev)
            rng = ou.make_rng(n, 7, 0.5)
            rng = {k: ({i: v.to(dev) for i, v in val.items()} if isinstance(val, dict) else val.to(dev)) for k, val in rng.items()}
            LOSSES["aw_triplet"].n_real = n // 2
            out = m(x, rng=rng)
            ld = out["loss_dict"]
            loss = LOSSES["cross_entropy"](out["cls_out"], tgt) + 0.1 * (ld["freq_mask"].mean() + ld["spat_mask"].mean()) \
                + 0.1 * sum(LOSSES["aw_triplet"](f, tgt) for f in ld["triplet"]) + out["rec"].square().mean()
            (loss * 64.0).backward()
            torch.cuda.synchronize()
            return ({k: out[k].detach().double() for k in ("cls_out", "rec")},
                    {k: p.grad.detach().double() / 64.0 for k, p in m.named_parameters() if p.grad is not None})
        finally:
            T._DW_TILED = saved
            LOSSES["aw_triplet"].n_real = None
    o0, g0 = run(False)
    o1, g1 = run(True)
    # half storage: 4 x what ONE rounding of parameters and inputs to fp16 does to the outputs of this step (0.9 - 1.3e-2,
    # tests/test_e_mixed_precision_gpu.py); observed 1.7 - 1.9e-2 (two valid roundings of the same half-storage step)
    otol, gtol = (5e-2, 5e-2) if half else (1e-4, 1e-3)
    for k in o0:
        e = float((o1[k] - o0[k]).norm() / o0[k].norm())
        assert within(f"{k}: tiled vs strip, relative L2", e, otol), (k, e)
    gscale = 1e-3 * max(float(v.norm()) for v in g0.values())
    rows = sorted(((float((g1[k] - g0[k]).norm() / (g0[k].norm() + gscale)), k) for k in g0), reverse=True)
    print("  worst gradients, tiled vs strip:", ", ".join(f"{e:.2e} {k}" for e, k in rows[:6]))
    if not half:
        assert within("parameter gradients: tiled vs strip, worst relative L2", rows[0][0], gtol), rows[:6]
    else:
        # a changed fp16 rounding early in the trunk is amplified like any perturbation of that size (tests/
        # test_e_mixed_precision_gpu.py measures the step's conditioning: one rounding of the parameters moves the gradients by
        # 5 % at the median, the cancelling scalar gates and the arg-max of the dynamic filters by tens of percent): distribution
        vals = sorted(e for e, k in rows if not k.endswith("_coef"))
        med, p90 = vals[len(vals) // 2], vals[int(0.9 * len(vals))]
        print(f"  half storage: median {med:.3g}, 90 % {p90:.3g}, worst {rows[0][0]:.3g}")
        ok = [within("parameter gradients: tiled vs strip, median relative L2 (4 x the one-rounding yardstick)", med, 0.19),
              within("parameter gradients: tiled vs strip, 90 % relative L2 (4 x the one-rounding yardstick)", p90, 0.21)]
        assert all(ok)
