"""GPU: the fused MBConv path (tape.mbconv_fused: deferred BatchNorm, csrc/fused.hip) against the operator-by-operator
path it replaces (tape.conv1x1 / batchnorm_act / sfconv_dw / squeeze_excite / residual, each tested against torch in
float64 by tests/test_kernels_gpu.py and against the reference's goldens by tests/test_model_gpu.py).

Both paths run the same parameters, inputs and dropout / drop-connect masks; the comparison is per tensor, relative to
the tensor's largest magnitude.  fp32 kernels in a different summation order, v_exp / v_rcp swish in the fused
kernels (1e-6): outputs 2e-4, activation gradients 1e-3, parameter gradients 3e-4 (+ the 2e-5 floor of test_model_gpu).
"""
import numpy as np
import pytest
import torch

from oracle import param_fill
from tests import oracle_util as ou

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def _run(dev, fused, sf, n, seed, running, debug=True):
    import unidefense_amd.model.unidefense as U
    from unidefense_amd.loss import LOSSES
    from unidefense_amd.model import load_model
    m = load_model("UDEB4")(extractor="efficientnet-b4", num_classes=2, drop_rate=0.5)
    param_fill.fill_module_(m, sf_coef=sf, fuse_coef=0.3)
    m = m.to(dev).train()
    x = param_fill.make_input(n, 256, seed).to(dev)
    tgt = param_fill.make_labels(n).to(dev)
    rng = ou.make_rng(n, 1, 0.5)
    rng = {k: ({i: v.to(dev) for i, v in val.items()} if isinstance(val, dict) else val.to(dev)) for k, val in rng.items()}
    saved = U._FUSED_MBCONV
    U._FUSED_MBCONV = fused
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
        U._FUSED_MBCONV = saved
        LOSSES["aw_triplet"].n_real = None
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    feats = {k: v.detach().clone() for k, v in m._debug_feats.items()} if debug else {}
    fgrads = {k: v.detach().clone() for k, v in m._debug_tape.captured.items() if v is not None} if debug else {}
    bufs = {k: v.detach().clone() for k, v in m.named_buffers()} if running else {}
    return loss.detach(), out, feats, grads, fgrads, bufs


@pytest.mark.parametrize("sf", [0.0, -10.0])
def test_fused_mbconv_equals_operator_path(sf):
    dev = _dev()
    l0, o0, f0, g0, fg0, b0 = _run(dev, False, sf, 4, 11, True)
    l1, o1, f1, g1, fg1, b1 = _run(dev, True, sf, 4, 11, True)
    assert _rel(l1, l0) < 1e-5
    bad = []
    for k in f0:
        e = _rel(f1[k], f0[k])
        if e > 2e-4:
            bad.append(("feat " + k, e))
    for k in fg0:
        e = _rel(fg1[k], fg0[k])
        if e > 1e-3:
            bad.append(("dfeat " + k, e))
    assert set(g0) == set(g1)
    worst = 0.0
    for k in g0:
        # the tolerance of tests/test_model_gpu.py: relative to the tensor's largest entry plus an absolute floor —
        # BN2's bias gradient is a sum that cancels to rounding noise (the next block's BatchNorm removes any
        # per-channel shift of its input), a purely relative bar is meaningless there
        d = float((g1[k].double() - g0[k].double()).abs().max())
        ref = float(g0[k].double().abs().max())
        # scalar gate gradients are global sums with heavy cancellation at sf_coef = -10 (sigmoid' = 4.5e-5)
        tol = (2e-3 if k.endswith("sf_coef") else 3e-4) * ref + 2e-5
        worst = max(worst, d / max(ref, 1e-30) if d > 2e-5 else 0.0)
        if d > tol:
            bad.append(("grad " + k, d, ref))
    print(f"worst parameter-gradient deviation {worst:.2e} over {len(g0)} tensors")
    # running statistics and batch counters move identically
    for k in b0:
        if b0[k].dtype.is_floating_point:
            e = _rel(b1[k], b0[k])
            if e > 1e-5:
                bad.append(("buffer " + k, e))
        else:
            assert torch.equal(b0[k], b1[k]), k
    assert not bad, bad[:20]


def test_skip_gradient_accumulated_in_place_equals_separate_add():
    """Without a debug watch the expand conv's data gradient is accumulated INTO the skip branch's gradient buffer
    (GEMM epilogue) instead of a separate add: same gradients."""
    dev = _dev()
    _, _, _, g0, _, _ = _run(dev, True, 0.0, 4, 3, False, debug=True)
    _, _, _, g1, _, _ = _run(dev, True, 0.0, 4, 3, False, debug=False)
    bad = [(k, float((g1[k] - g0[k]).abs().max())) for k in g0
           if float((g1[k] - g0[k]).abs().max()) > (2e-3 if k.endswith("sf_coef") else 1e-4) * float(g0[k].abs().max()) + 2e-5]
    assert not bad, bad[:10]
